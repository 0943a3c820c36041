# light round-end refresh: default bench line + kernel stats (WCSPH, DFSPH) + the other workloads' lines
TAG=${1:-r01j}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
for w in c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv; do timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --cpu-steps 0 > $OUT/bench_$w.json 2>/dev/null; done
for w in c3p_uniform_1.75M c2_dragon_bath; do timeout 300 python bench.py --solver dfsph --workload $w --steps 50 --warmup 10 --cpu-steps 0 > $OUT/bench_dfsph_$w.json 2>/dev/null; done
timeout 300 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --sweep > /dev/null 2> $OUT/sweep.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_wcsph -o prof -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof_wcsph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_dfsph -o prof -- python $R/bench.py --solver dfsph --steps 20 --warmup 5 --cpu-steps 0 > $OUT/rocprof_dfsph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/prof_wcsph -name "*.db" | head -1) $OUT/kernel_stats_c3p.txt > /dev/null
python tools/rocpd_summary.py $(find $OUT/prof_dfsph -name "*.db" | head -1) $OUT/kernel_stats_dfsph_c3p.txt > /dev/null
rm -rf $OUT/prof_wcsph $OUT/prof_dfsph
head -14 $OUT/kernel_stats_c3p.txt | cut -c1-160
