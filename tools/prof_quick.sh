R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o prof -- python $R/bench.py --steps 30 --warmup 5 --cpu-steps 0 > $R/gpurun_out/prof_q.log 2>&1
cd $R; python tools/rocpd_summary.py $(find gpurun_out/prof_q -name "*.db" | head -1) gpurun_out/prof_q.txt > /dev/null; head -8 gpurun_out/prof_q.txt | cut -c1-150; rm -rf gpurun_out/prof_q
