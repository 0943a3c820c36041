#!/bin/bash
# round 2, call h: the whole GPU suite on the new defaults, PMC passes -> pmc_traffic.json of THIS revision, kernel stats, bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02p
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 16 $OUT/pytest_gpu.log
cp gpurun_out/parity_curves.json $OUT/ 2>/dev/null; cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
bash tools/gpu_pmc.sh r02p > $OUT/pmc.log 2>&1; tail -n 3 $OUT/pmc.log
python tools/refresh_pmc.py gpurun_out/pmc_r02p $OUT/pmc_traffic.json > $OUT/refresh.log 2>&1; echo "refresh rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_r02p $OUT/pmc_summary.txt > /dev/null 2>&1
rm -rf gpurun_out/pmc_r02p
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_c3p.csv && head -n 12 "$f" | cut -c1-160
rm -rf $OUT/prof
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --settle 2000 > $OUT/bench_c3p_hydrostatic.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c1_dambreak_262k > $OUT/bench_c1_rest.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c1_dambreak_262k --settle 2500 > $OUT/bench_c1_developed.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c2_dragon_bath > $OUT/bench_c2.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c3_armadillo_equiv > $OUT/bench_c3.json 2>> $OUT/bench.err
for f in c3p_hydrostatic c1_rest c1_developed c2 c3; do python -c "import json;d=json.load(open('$OUT/bench_$f.json'));print('$f',d['ms_per_step'],d['steps_per_s_job'],d['breakdown_ms'])"; done
