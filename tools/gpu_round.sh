#!/bin/bash
# One GPU-box visit: parity tests, bench (+ variant sweep), rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
rocm-smi --showproductname 2>/dev/null | head -20 > $OUT/gpu.txt
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -x 2>&1 | tail -40 | tee $OUT/pytest_$TAG.log
echo "== bench"
timeout 900 python bench.py --steps 100 --warmup 10 --sweep > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
tail -5 $OUT/bench_$TAG.err; cat $OUT/bench_$TAG.json
echo "== rocprof"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o prof -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof_$TAG.log 2>&1
ls $OUT/prof_$TAG 2>/dev/null | head
find $OUT/prof_$TAG -name "*kernel_stats*" | head -1 | xargs -r head -30
