#!/bin/bash
# PMC passes over a short bench run (counters in their own runs, kernel-trace only).
# Usage: bash tools/gpu_pmc.sh <tag> [bench args...]
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --min-seconds 0 --settled-after 0 --with-bodies 0 $@"
run() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o p -- $BENCH > $OUT/$n.log 2>&1
  echo "pass $n rc=$?"
}
want() { case " ${PMC_PASSES:-sq1 sq2 sq3 fetch write tcc} " in *" $1 "*) return 0;; *) return 1;; esac; }
runw() { if want $1; then run "$@"; fi; }
runw sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
runw sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
runw sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
runw fetch FETCH_SIZE
runw write WRITE_SIZE
runw tcc TCC_HIT_sum TCC_MISS_sum
find $OUT -name "*counter_collection.csv" | head
