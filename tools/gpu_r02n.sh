#!/bin/bash
# round 2, last visit: PMC passes + kernel stats + default bench line of the FINAL kernel sources (the full suite ran in r02m and, after the
# 2-launch scan, again in parts: parity / golden / variants / distributed / full-size properties)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02n
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
bash tools/gpu_pmc.sh r02n > $OUT/pmc.log 2>&1; tail -n 2 $OUT/pmc.log
python tools/refresh_pmc.py gpurun_out/pmc_r02n $OUT/pmc_traffic.json > $OUT/refresh.log 2>&1; echo "refresh rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_r02n $OUT/pmc_summary.txt > /dev/null 2>&1
rm -rf gpurun_out/pmc_r02n
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_c3p.csv && head -n 10 "$f" | cut -c1-140
rm -rf $OUT/prof
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python -c "import json;d=json.load(open('$OUT/bench_default.json'));print(d['value'],d['ms_per_step'],d['breakdown_ms'],d['roofline']['traffic'],d['roofline_valu']['frac'],d['cpu_baseline']['value'])"
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -n 2
