#!/bin/bash
# round 2, call d: where the density sweep's time goes -- ablation table for kernel variants, rest and settled
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
rm -f $OUT/ablate.txt
for v in ${VARIANTS:-5 7}; do
  for s in 0 2000; do
    echo "== variant $v settle $s" >> $OUT/ablate.txt
    timeout 120 python bench.py --ablate --variant $v --settle $s --steps 20 --warmup 5 --cpu-steps 0 2>> $OUT/ablate.txt > /dev/null
  done
done
grep -v amdgpu.ids $OUT/ablate.txt
