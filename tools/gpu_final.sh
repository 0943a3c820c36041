#!/bin/bash
# Round-end measurement visit: smoke, full GPU suite, bench lines (all workloads, both solvers),
# rocprofv3 kernel stats and PMC passes.  Usage: bash tools/gpu_final.sh <tag>
TAG=${1:-r01h}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== bench default"; timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json
for w in c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv; do
  timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --cpu-steps 0 > $OUT/bench_$w.json 2>/dev/null
done
timeout 300 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --sweep > /dev/null 2> $OUT/sweep.txt
for w in c3p_uniform_1.75M c2_dragon_bath; do
  timeout 300 python bench.py --solver dfsph --workload $w --steps 50 --warmup 10 --cpu-steps 0 > $OUT/bench_dfsph_$w.json 2>/dev/null
done
echo "== rocprof stats"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_wcsph -o prof -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof_wcsph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_dfsph -o prof -- python $R/bench.py --solver dfsph --steps 20 --warmup 5 --cpu-steps 0 > $OUT/rocprof_dfsph.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/prof_wcsph -name "*.db" | head -1) $OUT/kernel_stats_c3p.txt > /dev/null
python tools/rocpd_summary.py $(find $OUT/prof_dfsph -name "*.db" | head -1) $OUT/kernel_stats_dfsph_c3p.txt > /dev/null
head -12 $OUT/kernel_stats_c3p.txt | cut -c1-170
echo "== pmc"
GRAFT_REPO_ROOT=$R bash tools/gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmc_$TAG $OUT/pmc_c3p.txt > /dev/null
rm -rf $OUT/prof_wcsph $OUT/prof_dfsph $R/gpurun_out/pmc_$TAG
ls $OUT
