R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
for w in c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o prof -- python $R/bench.py --workload $w --steps 50 --warmup 5 --cpu-steps 0 > $R/gpurun_out/prof_$w.log 2>&1
  python $R/tools/rocpd_summary.py $(find $R/gpurun_out/prof_$w -name "*.db" | head -1) $R/gpurun_out/kernel_stats_$w.txt > /dev/null
  rm -rf $R/gpurun_out/prof_$w
  head -7 $R/gpurun_out/kernel_stats_$w.txt | cut -c1-160
done
