"""What a partitioned handle holds: stored entries of every level-0 / level-1 operator and the pool's bytes, whole vs rank 0 of P (one process;
no exchange needed for the set-up).  usage: python scripts/partition_probe.py [P]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gravo_mg_amd import cabi
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, mass, lhs, rhs = bench.build_workload(1732, 1732)
def info(e, tag):
    out = {}
    for k in (0, 1):
        for which, name in ((0, "A"), (3, "P"), (4, "R")):
            i = (cabi.C.c_int64 * 4)()
            e._chk(cabi.lib().gmg_debug_sell_info(e._h, k, which, i))
            out[f"l{k}{name}"] = int(i[2])
    print(tag, out, "device_bytes_now MB", round(e.timing("device_bytes_now") / 1e6), "after set-up", round(e.timing("device_bytes") / 1e6), "peak", round(e.timing("device_bytes_peak") / 1e6),
          "set_system_ms", round(e.timing("setup_total"), 1), flush=True)
whole = cabi.Engine(row_align=64 * P, block_fine=0)
whole.use_hierarchy(H); whole.set_mass(mass); whole.set_system(lhs)
info(whole, "whole ")
whole.close(); del whole
for prep in (True,):
    part = cabi.Engine(row_align=64 * P, block_fine=0, prepare_structure=prep)
    part.dist_partition(0, P)
    part.use_hierarchy(H); part.set_mass(mass); part.set_system(lhs)
    info(part, f"rank0/{P} prep={prep}")
    marks = {}
    for m in ["pattern_key", "upload_A0"] + [f"rap_l{k}" for k in range(1, 6)] + [f"ordering_ready_l{k}" for k in range(6)] + ["device_layout", "tasks_joined", "factor_joined", "mass_done"]:
        try:
            marks[m] = round(part.timing("t_" + m), 2)
        except Exception:
            pass
    print("    marks", sorted(marks.items(), key=lambda kv: kv[1]))
    print("    parked after set-up MB: see device_bytes_now vs hipMemGetInfo")
    for key in ("dist_plan_ms", "dist_plan_cached", "setup_ordering_cached", "reduction", "setup_device_layout"):
        try:
            print("   ", key, part.timing(key))
        except Exception as e:
            print("   ", key, "-")
    part.close(); del part
