#!/bin/bash
# round-6 visit f: the final evidence of the round on the final tree -- the whole GPU suite, PMC passes (the kernel sources changed: the
# fingerprint in profiles/pmc_traffic*.json must be this tree's), kernel traces, the default line + the other workloads, the real RCCL
# path as one rank, and the A/B of the DFSPH list writer's emission (group-sorted, new in r06, against the baseline emission)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06f
mkdir -p $OUT
cd $R
bash tools/gpu_round.sh r06f tests bench kstats pmc pmcdf native dfgaps
for rep in 1 2; do
  for v in 25 24; do
    timeout 300 python bench.py --cpu-steps 0 --solver dfsph --steps 30 --warmup 3 --variant $v > $OUT/bench_dfsph_v${v}_$rep.json 2>> $OUT/bench_df.err; echo "dfsph variant=$v rep=$rep rc=$?"
    python -c "import json;d=json.load(open('$OUT/bench_dfsph_v${v}_$rep.json'));print('dfsph variant=$v', d['value'], d['ms_per_step'], d['dfsph']['ms_per_sweep'], d['dfsph']['neighbour_sweeps_per_step'])"
  done
done
