#!/bin/bash
# round-6 visit d: the slab driver after (a) the layer offsets folded into the sort's place kernel, (b) the pure-fluid density instance
# vouched for by the scene file: slab tests, world = 1 overhead against the plain loop, kernel trace + gaps of the slab step
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06d
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_distributed.py tests/test_gpu_c4.py -m gpu -q -x --durations=12 > $OUT/pytest_slabs.log 2>&1; echo "slab tests rc=$?"
tail -n 8 $OUT/pytest_slabs.log
timeout 300 python tools/slab_overhead.py 2>&1 | grep -v "amdgpu.ids" > $OUT/slab_overhead_world1.txt; cat $OUT/slab_overhead_world1.txt | head -n 6
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profslab -o prof --output-format csv -- python $R/tools/slab_overhead.py > $OUT/slab_overhead_world1_traced.txt 2>&1 )
f=$(find $OUT/profslab -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_slab_world1.csv && python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_slab_world1.csv")))
for r in rows[:12]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
t=$(find $OUT/profslab -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/ktrace_gaps.py "$t" 0.5 at=0.45 at=0.8 > $OUT/slab_gaps.txt 2>&1 && head -n 100 $OUT/slab_gaps.txt
rm -rf $OUT/profslab
