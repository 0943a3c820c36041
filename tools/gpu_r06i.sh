#!/bin/bash
# round-6 visit i: SPH_OPT_BRICK_ORIGIN (column groups start at cell 1) -- variant + sort + parity tests, then the A/B on the headline
# line (rest + settled), C1, C2 and the DFSPH line, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06i
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x --durations=5 > $OUT/pytest_part.log 2>&1; echo "tests rc=$?"
tail -n 8 $OUT/pytest_part.log
for rep in 1 2 3; do
  for o in 1 0; do
    timeout 300 python bench.py --cpu-steps 0 --with-bodies 0 --brick-origin $o > $OUT/bench_default_o${o}_$rep.json 2>> $OUT/bench.err; echo "bench origin=$o rep=$rep rc=$?"
    python -c "import json;d=json.load(open('$OUT/bench_default_o${o}_$rep.json'));print('origin=$o', d['value'], d['ms_per_step'], d['breakdown_ms'], 'settled', d['settled']['value'], d['settled']['ms_per_step'], d['settled']['breakdown_ms'])"
  done
done
for rep in 1 2; do
  for o in 1 0; do
    for w in c1_dambreak_262k c2_dragon_bath; do
      timeout 200 python bench.py --steps 200 --warmup 10 --cpu-steps 0 --workload $w --settled-after 0 --min-seconds 0 --brick-origin $o > $OUT/bench_${w}_o${o}_$rep.json 2>> $OUT/bench.err
      python -c "import json;d=json.load(open('$OUT/bench_${w}_o${o}_$rep.json'));print('$w origin=$o', d['value'], d['ms_per_step'], d['breakdown_ms'])"
    done
    timeout 300 python bench.py --cpu-steps 0 --solver dfsph --steps 30 --warmup 3 --brick-origin $o > $OUT/bench_dfsph_o${o}_$rep.json 2>> $OUT/bench.err
    python -c "import json;d=json.load(open('$OUT/bench_dfsph_o${o}_$rep.json'));print('dfsph origin=$o', d['value'], d['ms_per_step'], d['dfsph']['ms_per_sweep'])"
  done
done
