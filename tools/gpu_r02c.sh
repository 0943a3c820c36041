#!/bin/bash
# round 2, call c: parity of every SPH_OPT_KERNEL_VARIANT instance + their A/B timing table (rest lattice, settled flow)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_variants.py -m gpu -q -x --durations=5 > $OUT/pytest_variants.log 2>&1; echo "variants pytest rc=$?"
tail -n 15 $OUT/pytest_variants.log
timeout 900 python tools/variant_sweep.py --out $OUT/variants.json > $OUT/variant_sweep.log 2>&1; echo "sweep rc=$?"
cat $OUT/variant_sweep.log | tail -n 40
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "kernel_by_kernel or trajectory or edge_cases or force_paths" > $OUT/pytest_parity.log 2>&1; echo "parity pytest rc=$?"
tail -n 8 $OUT/pytest_parity.log
