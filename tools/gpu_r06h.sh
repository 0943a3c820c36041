#!/bin/bash
# round-6 visit h: the final evidence on the final tree (csrc changed after r06f: the fused DFSPH error reduction) + the DFSPH order probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06h
mkdir -p $OUT
cd $R
timeout 600 python tools/df_order_probe.py 63 2> $OUT/df_order_probe.err | grep '^{' > $OUT/df_order_probe.json; cat $OUT/df_order_probe.json
bash tools/gpu_round.sh r06h tests bench kstats pmc pmcdf native dfgaps
