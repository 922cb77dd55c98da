"""CPU restatement of the reference hot path -- TEST INFRASTRUCTURE ONLY (see gravomg_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
