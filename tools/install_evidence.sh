#!/bin/bash
# Copy what a `tools/gpu_round.sh <tag> tests bench kstats pmc pmcdf native dfgaps` visit merged into gpurun_out/<tag>/ into profiles/
# under the names profiles/README.md lists (<tag>_*), and the two PMC files under their fixed names.  Usage: bash tools/install_evidence.sh r06l
T=$1
S=gpurun_out/$T
P=profiles
[ -d "$S" ] || { echo "no $S"; exit 1; }
cp $S/pmc_traffic.json $P/pmc_traffic.json; cp $S/pmc_traffic_dfsph.json $P/pmc_traffic_dfsph.json
cp $S/pmc_brief.json $P/${T}_pmc_brief.json; cp $S/pmc_dfsph_brief.json $P/${T}_pmc_dfsph_brief.json
for f in bench_default bench_driver_args bench_c4_dambreak_one_gpu parity_curves parity_errors golden_errors variant_errors long_bodies big_fixture_cell_id_mismatches; do cp $S/$f.json $P/${T}_$f.json; done
python - "$S" "$P/${T}_bench_other_workloads.json" <<'PY'
import json, sys
out = {f: json.load(open(f"{sys.argv[1]}/bench_{f}.json")) for f in ["c1_dambreak_262k", "c1_developed", "c2_dragon_bath", "c3_armadillo_equiv", "dfsph_c3p"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
cp $S/kernel_stats_1.csv $P/${T}_kernel_stats_c3p_rest.csv; cp $S/kernel_stats_2.csv $P/${T}_kernel_stats_c3p_settle2000_whole_run.csv
cp $S/kernel_tail_1.txt $P/${T}_kernel_tail_c3p_rest.txt; cp $S/kernel_tail_2.txt $P/${T}_kernel_tail_c3p_settled.txt
cp $S/kernel_stats_dfsph.csv $P/${T}_kernel_stats_dfsph.csv; cp $S/dfsph_gaps.txt $P/${T}_dfsph_gaps.txt
cp $S/native_rccl_one_rank_kernel_stats.csv $P/${T}_native_rccl_one_rank_kernel_stats.csv; cp $S/slab_overhead_world1.txt $P/${T}_slab_overhead_world1.txt
tail -n 22 $S/pytest_gpu.log > $P/${T}_pytest_gpu_tail.txt
python - <<'PY'
import json
from sph_taichi_amd import build
fp = build._fingerprint()
for f in ("profiles/pmc_traffic.json", "profiles/pmc_traffic_dfsph.json"):
    print(f, "fingerprint matches the tree:", json.load(open(f))["kernel_fingerprint"] == fp)
PY
