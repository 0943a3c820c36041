#!/bin/bash
# round-5 visit i: the pure-fluid instance of the density sweep (m_V_j = m_V0 from a register in contexts without solids; bit-identical): A/B on the
# same box (SPH_DISABLE_PURE_FLUID = the general instance), then the GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05i2
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for v in on off; do
  if [ $v = off ]; then export SPH_DISABLE_PURE_FLUID=1; else unset SPH_DISABLE_PURE_FLUID; fi
  timeout 200 python bench.py --steps 200 --warmup 20 --cpu-steps 0 --with-bodies 0 > $OUT/bench_pure_${v}_$rep.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_pure_${v}_$rep.json'));print('pure $v', d['value'], d['breakdown_ms']['neighbour'], 'settled', d['settled']['value'], d['settled']['breakdown_ms']['neighbour'])"
done
done
unset SPH_DISABLE_PURE_FLUID
SPH_TEST_EVIDENCE_DIR=$OUT timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"; tail -n 6 $OUT/pytest_gpu.log
