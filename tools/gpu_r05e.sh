#!/bin/bash
# round-5 visit e: the matrix-pipe filter variant (SPH_VAR_MFMA = 32; 57 = default | MFMA): parity of the variant tests, then A/B timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05e
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_variants.py -m gpu -q -x > $OUT/pytest_variants.log 2>&1; echo "variant tests rc=$?"; tail -n 15 $OUT/pytest_variants.log
for v in 25 57; do
  timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --with-bodies 0 --variant $v --min-seconds 0 > $OUT/bench_v$v.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_v$v.json'));print('variant $v rest', d['value'], d['breakdown_ms'], 'settled', d['settled']['value'], d['settled']['breakdown_ms'])"
done
