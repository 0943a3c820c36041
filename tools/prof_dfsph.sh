R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dfsph -o prof -- python $R/bench.py --solver dfsph --steps 20 --warmup 5 --cpu-steps 0 > $R/gpurun_out/prof_dfsph.log 2>&1
cd $R; f=$(find gpurun_out/prof_dfsph -name "*.db" | head -1); python tools/rocpd_summary.py $f gpurun_out/prof_dfsph_summary.txt; head -30 gpurun_out/prof_dfsph_summary.txt | cut -c1-200; tail -1 gpurun_out/prof_dfsph.log | cut -c1-600
