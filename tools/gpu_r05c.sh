#!/bin/bash
# round-5 visit c: the two fixed tests, kernel trace of the default line (rest) and of the DFSPH line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "turned_body or c4_object_hangs or flat_body" > $OUT/pytest_sel.log 2>&1; echo "selected tests rc=$?"; tail -n 5 $OUT/pytest_sel.log
bash tools/gpu_kstats.sh r05c "--steps 60 --warmup 5 --min-seconds 0 --settled-after 0"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profd -o prof --output-format csv -- python $R/bench.py --cpu-steps 0 --steps 30 --warmup 3 --solver dfsph > $OUT/rocprof_dfsph.log 2>&1 )
f=$(find $OUT/profd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_dfsph.csv && head -n 14 $OUT/kernel_stats_dfsph.csv | cut -c1-160
t=$(find $OUT/profd -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/ktrace_gaps.py "$t" > $OUT/dfsph_gaps.txt 2>&1 && tail -n 12 $OUT/dfsph_gaps.txt
grep -h '^{"metric"' $OUT/rocprof_dfsph.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('dfsph line under the profiler:', d['value'], d['ms_per_step'], d['dfsph'])"
rm -rf $OUT/profd
