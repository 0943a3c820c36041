#!/bin/bash
# round-6 visit g: SPH_OPT_DF_FUSE_ERROR -- the DFSPH tests (golden fixtures with exact iteration counts, the new A/B test) and the A/B of the
# DFSPH line, alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06g
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "dfsph or DFSPH or df_" --durations=5 > $OUT/pytest_dfsph.log 2>&1; echo "dfsph tests rc=$?"
tail -n 10 $OUT/pytest_dfsph.log
for rep in 1 2 3 4; do
  for f in 1 0; do
    timeout 300 python bench.py --cpu-steps 0 --solver dfsph --steps 60 --warmup 3 --df-fuse-error $f > $OUT/bench_dfsph_fuse${f}_$rep.json 2>> $OUT/bench_df.err; echo "dfsph fuse=$f rep=$rep rc=$?"
    python -c "import json;d=json.load(open('$OUT/bench_dfsph_fuse${f}_$rep.json'));print('dfsph fuse=$f', d['value'], d['ms_per_step'], d['dfsph']['ms_per_sweep'], d['dfsph']['neighbour_sweeps_per_step'], d['dfsph'].get('iterations'))"
  done
done
bash tools/gpu_round.sh r06g dfgaps
python - <<'P'
import csv
rows=list(csv.DictReader(open("gpurun_out/r06g/kernel_stats_dfsph.csv")))
for r in rows[:16]: print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us")
P
