#!/bin/bash
# rocprofv3 kernel-trace statistics of bench.py runs.  Usage: bash tools/gpu_kstats.sh <tag> "<bench args>" ["<bench args>" ...]
# Writes gpurun_out/<tag>/kernel_stats_<i>.csv (whole run) and kernel_tail_<i>.txt (mean of each kernel's last 50 dispatches:
# the timed steps, without the settling steps before them).
TAG=${1:-ks}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for ARGS in "$@"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof$i -o prof --output-format csv -- python $R/bench.py --cpu-steps 0 --with-bodies 0 $ARGS > $OUT/rocprof$i.log 2>&1 )
  f=$(find $OUT/prof$i -name "*kernel_stats.csv" | head -1)
  t=$(find $OUT/prof$i -name "*kernel_trace.csv" | head -1)
  echo "== $ARGS"
  grep -h '^{"metric"' $OUT/rocprof$i.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench line of the profiled run:', d['ms_per_step'], d['breakdown_ms'])"
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$i.csv
  [ -n "$t" ] && python $R/tools/ktrace_tail.py "$t" 50 > $OUT/kernel_tail_$i.txt && head -n 9 $OUT/kernel_tail_$i.txt
  rm -rf $OUT/prof$i
done
