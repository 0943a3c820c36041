#!/bin/bash
# round 3, call c: lean records (eos2, lazy aux, acceleration stored on the last step only) + the faster k_brick_list
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=3 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 8 $OUT/pytest_gpu.log
bash tools/gpu_kstats.sh r03c "--steps 60 --warmup 5" "--steps 60 --warmup 5 --settle 2000"
timeout 300 python tools/variant_sweep.py --variants 189,285 --shapes 0 --steps 60 --settled-steps 80 --out $OUT/variants.json > $OUT/variants.log 2>&1
grep -v "^$" $OUT/variants.log | tail -n 7
