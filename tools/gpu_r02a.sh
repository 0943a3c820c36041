#!/bin/bash
# round 2, call a: VALU/LDS issue-rate microbenchmark + the deep parity cases
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
nproc > $OUT/nproc.txt
timeout 300 tools/bin/ubench_valu > $OUT/ubench_valu.txt 2>&1; echo "ubench rc=$?"
tail -n 80 $OUT/ubench_valu.txt
rm -f gpurun_out/parity_curves.json
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x --durations=0 \
   -k "impact or onto_dragon or c1_ or c3p_headline" > $OUT/pytest_deep.log 2>&1; echo "pytest rc=$?"
tail -n 40 $OUT/pytest_deep.log
cp gpurun_out/parity_curves.json $OUT/ 2>/dev/null
cat $OUT/parity_curves.json
