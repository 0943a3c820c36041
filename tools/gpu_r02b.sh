#!/bin/bash
# round 2, call b: second opcode table, developed-flow bench lines, C2 timing split, remaining deep parity cases
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 300 tools/bin/ubench_valu2 > $OUT/ubench_valu2.txt 2>&1; echo "ubench2 rc=$?"
cat $OUT/ubench_valu2.txt
B="python bench.py --steps 100 --warmup 10 --cpu-steps 0"
timeout 300 $B > $OUT/bench_c3p_rest.json 2> $OUT/bench_c3p_rest.err; tail -c 1500 $OUT/bench_c3p_rest.json
timeout 300 $B --settle 2000 > $OUT/bench_c3p_hydrostatic.json 2>> $OUT/bench.err; tail -c 900 $OUT/bench_c3p_hydrostatic.json
timeout 300 $B --workload c3p_slosh_1.75M --settle 1500 > $OUT/bench_c3p_slosh.json 2>> $OUT/bench.err; tail -c 900 $OUT/bench_c3p_slosh.json
timeout 300 $B --workload c1_dambreak_262k > $OUT/bench_c1_rest.json 2>> $OUT/bench.err; tail -c 900 $OUT/bench_c1_rest.json
timeout 300 $B --workload c1_dambreak_262k --settle 2500 > $OUT/bench_c1_developed.json 2>> $OUT/bench.err; tail -c 900 $OUT/bench_c1_developed.json
timeout 300 $B --workload c2_dragon_bath --settle 300 > $OUT/bench_c2_after_impact.json 2>> $OUT/bench.err; tail -c 900 $OUT/bench_c2_after_impact.json
tail -5 $OUT/bench.err
timeout 600 python tools/time_c2.py > $OUT/time_c2.txt 2>&1; cat $OUT/time_c2.txt
rm -f gpurun_out/parity_curves.json
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --durations=0 \
   -k "onto_dragon or c1_ or c3p_headline" > $OUT/pytest_deep.log 2>&1; echo "pytest rc=$?"
tail -n 30 $OUT/pytest_deep.log
cp gpurun_out/parity_curves.json $OUT/ 2>/dev/null
