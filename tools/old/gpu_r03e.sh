#!/bin/bash
# round 3, call e: the native RCCL exchange (sph_comm.hip) -- suite, one-rank kernel trace, world = 1 overhead
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=3 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 12 $OUT/pytest_gpu.log
timeout 300 python tools/slab_overhead.py > $OUT/slab_overhead_world1.txt 2>&1; cat $OUT/slab_overhead_world1.txt | grep -v amdgpu.ids
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof --output-format csv -- python -m pytest $R/tests/test_distributed.py -q -k native_rccl > $OUT/rocprof_native.log 2>&1 )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/native_rccl_one_rank_kernel_stats.csv && grep -i "nccl\|rccl" "$f" | cut -c1-160
rm -rf $OUT/prof
