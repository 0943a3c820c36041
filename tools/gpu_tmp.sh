#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in 0 29; do
rm -f gpurun_out/parity_curves.json
SPH_KERNEL_VARIANT=$v timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "floor_impact" 2>&1 | grep -E "^E +assert|passed|failed" | head -3
python - <<PY
import json
d=json.load(open("gpurun_out/parity_curves.json"))["c2_dragon_bath"]
print("variant $v", {k: d[k] for k in ("rel_l2_x","rel_l2_density","rel_l2_v")})
PY
done
