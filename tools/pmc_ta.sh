# TA / TCP counters of the two sweeps: bash tools/pmc_ta.sh [extra bench.py arguments, e.g. --settle 2000]
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
OUT=$R/gpurun_out/pmc_ta; rm -rf $OUT; mkdir -p $OUT
BENCH="python $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --min-seconds 0 --settled-after 0 $*"
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- $BENCH > $OUT/$n.log 2>&1; echo "pass $n rc=$?"; }
run ta1 TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
run ta2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES
cd $R; python tools/pmc_summary.py $OUT $OUT/summary.txt > /dev/null; echo "# bench arguments: $*"; grep -A20 "k_gather_brick<14" $OUT/summary.txt | head -24; grep -A20 "k_gather_brick<3" $OUT/summary.txt | head -24
