#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_variants.py -m gpu -q -x 2>&1 | tail -n 4
timeout 300 python tools/variant_sweep.py --variants 5,29,61 --out $OUT/variants_force2.json 2>&1 | grep -v amdgpu.ids
for v in 5 29; do
SPH_KERNEL_VARIANT=$v timeout 200 python bench.py --solver dfsph --steps 30 --warmup 5 --cpu-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dfsph variant $v', d['ms_per_step'], d['breakdown_ms'], d['dfsph'])"
done
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "dfsph" 2>&1 | tail -n 3
