#!/bin/bash
# round-6 visit b: (1) the GPU suite with the tightened bounds ASSERTED; (2) A/B of the brick column records (SPH_OPT_BRICK_RECORDS) on the
# headline line (rest + settled) and on the DFSPH line, alternating on the same box; (3) TA / TCP counters of the force sweep, rest and
# settled (VERDICT r05 "next" #5); (4) kernel trace of the slab driver at world = 1 against the plain loop
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06b
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SPH_TEST_EVIDENCE_DIR=$OUT/evidence timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 16 $OUT/pytest_gpu.log
for rep in 1 2; do
  for br in 1 0; do
    timeout 300 python bench.py --cpu-steps 0 --with-bodies 0 --brick-records $br > $OUT/bench_default_br${br}_$rep.json 2> $OUT/bench_br.err; echo "bench br=$br rep=$rep rc=$?"
    python -c "import json;d=json.load(open('$OUT/bench_default_br${br}_$rep.json'));print('br=$br', d['value'], d['ms_per_step'], d['breakdown_ms'], 'settled', d['settled']['value'], d['settled']['ms_per_step'], d['settled']['breakdown_ms'])"
  done
done
for br in 1 0 1 0; do
  timeout 300 python bench.py --cpu-steps 0 --solver dfsph --steps 30 --warmup 3 --brick-records $br > $OUT/bench_dfsph_br${br}.json 2> $OUT/bench_df.err; echo "dfsph br=$br rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_dfsph_br${br}.json'));print('dfsph br=$br', d['value'], d['ms_per_step'], d['dfsph']['ms_per_sweep'], d['dfsph']['neighbour_sweeps_per_step'])"
done
GRAFT_REPO_ROOT=$R bash tools/pmc_ta.sh --with-bodies 0 > $OUT/pmc_ta_rest.txt 2>&1; cp gpurun_out/pmc_ta/summary.txt $OUT/pmc_ta_rest_summary.txt 2>/dev/null; tail -n 50 $OUT/pmc_ta_rest.txt | head -n 60
GRAFT_REPO_ROOT=$R bash tools/pmc_ta.sh --with-bodies 0 --settle 2000 > $OUT/pmc_ta_settled.txt 2>&1; cp gpurun_out/pmc_ta/summary.txt $OUT/pmc_ta_settled_summary.txt 2>/dev/null; tail -n 50 $OUT/pmc_ta_settled.txt | head -n 60
rm -rf gpurun_out/pmc_ta
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profslab -o prof --output-format csv -- python $R/tools/slab_overhead.py > $OUT/slab_overhead_world1.txt 2>&1 )
f=$(find $OUT/profslab -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_slab_world1.csv && cut -c1-150 $OUT/kernel_stats_slab_world1.csv | head -n 24
rm -rf $OUT/profslab
grep -v amdgpu.ids $OUT/slab_overhead_world1.txt | head -n 8
