# PMC of the level-1 sweep (and, for comparison, of the fine-level colour kernel): one rocprofv3 pass per counter group
O=$GRAFT_REPO_ROOT/gpurun_out/r04c; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pmc() { N=$1; shift
  timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$N -- python $R/scripts/l1_sweep_min.py 1 > $O/pmc_$N.log 2>&1
  python $R/scripts/pmc_summary.py $(find $O/pmc_$N -name "*counter_collection.csv" | head -1) > $O/pmc_$N.txt 2>&1; rm -rf $O/pmc_$N; }
pmc C TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pmc D TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pmc D2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum
pmc E TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum
pmc F TCC_BUSY_avr TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pmc F2 TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pmc G SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE GRBM_TA_BUSY
grep -h "gs_block_ep<double, 1>\|gs_color<double, 1, 2>" $O/pmc_[C-G]*.txt | cut -c1-130
tail -3 $O/pmc_C.log
