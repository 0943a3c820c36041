#!/bin/bash
# round-6 visit c: the force sweep with its second neighbour record staged in LDS (SPH_VAR_GAT_LDS = 2: tile 1,408, 3 workgroups / CU;
# SPH_VAR_GAT_LDS4 = 4: tile 1,200, 4 / CU) against the default (25) and against the run-ordered lists of variant 24, rest and settled,
# alternating on one box; the variant tests; TA / TCP counters of the winner
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
SPH_TEST_EVIDENCE_DIR=$OUT/evidence timeout 900 python -m pytest tests/test_gpu_variants.py -m gpu -q > $OUT/pytest_variants.log 2>&1; echo "variant tests rc=$?"
tail -n 8 $OUT/pytest_variants.log
for rep in 1 2; do
  for v in 25 27 29 24; do
    timeout 300 python bench.py --cpu-steps 0 --with-bodies 0 --variant $v > $OUT/bench_v${v}_$rep.json 2> $OUT/bench_v.err; echo "bench v=$v rep=$rep rc=$?"
    python -c "import json;d=json.load(open('$OUT/bench_v${v}_$rep.json'));print('v=$v', d['value'], d['ms_per_step'], d['breakdown_ms'], 'settled', d['settled']['value'], d['settled']['ms_per_step'], d['settled']['breakdown_ms'], d['settled']['neighbourhood'])"
  done
done
