#!/bin/bash
# round 2, call f: whole GPU suite, slab overhead after the world=1 shortcuts, C4 in its own geometry (one context + 8 logical slabs with re-cut)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 22 $OUT/pytest_gpu.log
timeout 300 python tools/slab_overhead.py > $OUT/slab_overhead.txt 2>&1; grep -v amdgpu.ids $OUT/slab_overhead.txt
timeout 900 python tools/c4_geometry.py --steps 200 --recut-every 10 --out $OUT/c4_geometry.json > $OUT/c4.log 2>&1; echo "c4 rc=$?"; grep -v amdgpu.ids $OUT/c4.log | tail -n 8
