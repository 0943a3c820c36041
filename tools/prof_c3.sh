R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c3 -o prof -- python $R/bench.py --workload c3_armadillo_equiv --steps 50 --warmup 5 --cpu-steps 0 > $R/gpurun_out/prof_c3.log 2>&1
cd $R; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shape_matched" --timeout=200 2>&1 | tail -3
