#!/bin/bash
# r01m: default bench line, kernel stats, and the one PMC pass that shows the LDS bank-conflict share of the filter
TAG=r01m; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_wcsph -o prof -- python $R/bench.py --steps 50 --warmup 5 --cpu-steps 0 > $OUT/rocprof_wcsph.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_$TAG/sq2 -o p -- python $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 > $OUT/pmc_sq2.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/prof_wcsph -name "*.db" | head -1) $OUT/kernel_stats_c3p.txt > /dev/null
python tools/pmc_summary.py $R/gpurun_out/pmc_$TAG $OUT/pmc_lds_c3p.txt > /dev/null
rm -rf $OUT/prof_wcsph $R/gpurun_out/pmc_$TAG
head -6 $OUT/kernel_stats_c3p.txt | cut -c1-160; head -16 $OUT/pmc_lds_c3p.txt | cut -c1-120
