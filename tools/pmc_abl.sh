R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_abl11; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python $R/bench.py --steps 10 --warmup 100 --cpu-steps 0 --ablate-mask 11"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- $B > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/b -o p -- $B > $OUT/b.log 2>&1
echo done
