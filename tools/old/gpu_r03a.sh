#!/bin/bash
# round 3, call a: adaptive-height bricks (k_brick_list) + the hit-mask ring (SPH_VAR_RING) -- parity suite, then the
# A/B table partition x variant at rest and settled, settled positions for the CPU models, section ablations
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 12 $OUT/pytest_gpu.log
timeout 600 python tools/variant_sweep.py --variants 189,285 --shapes 1,0 --steps 60 --settled-steps 80 --dump-settled $OUT/settled_x.npy --out $OUT/variants_partition.json > $OUT/variants.log 2>&1; echo "sweep rc=$?"
grep -v "^$" $OUT/variants.log | tail -n 14
for v in 189 285; do
  timeout 200 python bench.py --steps 40 --warmup 5 --cpu-steps 0 --variant $v --ablate > $OUT/ablate_rest_$v.json 2> $OUT/ablate_rest_$v.txt
  timeout 200 python bench.py --steps 40 --warmup 5 --cpu-steps 0 --variant $v --settle 2000 --ablate > $OUT/ablate_settled_$v.json 2> $OUT/ablate_settled_$v.txt
done
grep -h ablate $OUT/ablate_rest_285.txt $OUT/ablate_settled_285.txt | head -30
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c1_dambreak_262k --settle 2500 > $OUT/bench_c1_developed.json 2>> $OUT/bench.err
timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c2_dragon_bath > $OUT/bench_c2.json 2>> $OUT/bench.err
for f in c1_developed c2; do python -c "import json;d=json.load(open('$OUT/bench_$f.json'));print('$f',d['ms_per_step'],d['breakdown_ms'])"; done
