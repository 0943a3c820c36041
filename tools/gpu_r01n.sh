#!/bin/bash
# r01n (round end): full GPU suite, smoke, default bench line, the other workloads' lines, WCSPH kernel stats
TAG=r01n; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
timeout 400 python -m pytest tests -m gpu -q --timeout=300 2>&1 | tail -2 | tee $OUT/pytest.log
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
for w in c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv; do timeout 120 python bench.py --workload $w --steps 200 --warmup 20 --cpu-steps 0 > $OUT/bench_$w.json 2>/dev/null; done
for w in c3p_uniform_1.75M c2_dragon_bath; do timeout 120 python bench.py --solver dfsph --workload $w --steps 50 --warmup 10 --cpu-steps 0 > $OUT/bench_dfsph_$w.json 2>/dev/null; done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_wcsph -o prof -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof_wcsph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/prof_wcsph -name "*.db" | head -1) $OUT/kernel_stats_c3p.txt > /dev/null
rm -rf $OUT/prof_wcsph
head -8 $OUT/kernel_stats_c3p.txt | cut -c1-160
