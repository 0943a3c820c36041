#!/bin/bash
# round 2, call e: RCCL one-rank path (+ its kernel trace), conservation guard, add_cube/add_particles, slab overhead, new bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_distributed.py -m gpu -q -x -k "rccl or conservation" > $OUT/pytest_rccl.log 2>&1; echo "rccl pytest rc=$?"
tail -n 12 $OUT/pytest_rccl.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "add_cube" > $OUT/pytest_add.log 2>&1; echo "add pytest rc=$?"
tail -n 5 $OUT/pytest_add.log
# RCCL kernels in a kernel trace of the same worker
python - <<'PY'
import json, sys, os
sys.path.insert(0, "tests")
import scenes
sd = scenes.fluid_with_rigid_bodies("/tmp/cube_rccl.obj")
json.dump(sd, open("/tmp/scene_rccl.json", "w"))
PY
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/rccl_trace -o t --output-format csv -- python $R/tests/slab_worker.py rccl1 0 1 29517 /tmp/rccl_res.npz /tmp/scene_rccl.json 12 > $OUT/rccl_trace.log 2>&1 ); echo "rccl trace rc=$?"
find $OUT/rccl_trace -name "*kernel_stats.csv" | head -2
f=$(find $OUT/rccl_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 40 "$f" | cut -c1-200
timeout 300 python tools/slab_overhead.py > $OUT/slab_overhead.txt 2>&1; cat $OUT/slab_overhead.txt | grep -v amdgpu.ids
timeout 300 python bench.py --steps 100 --warmup 10 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 2500 $OUT/bench_default.json
