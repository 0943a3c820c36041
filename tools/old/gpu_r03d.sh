#!/bin/bash
# round 3, call d: force targets from the LDS tile, bench.py's two-state line
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=3 > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
tail -n 6 $OUT/pytest_gpu.log
( time timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
python - <<'P'
import json
d=json.load(open("gpurun_out/r03d/bench_default.json"))
print("rest", d["value"], d["ms_per_step"], d["reps"], d["timed_seconds"], d["first_rep"], d["breakdown_ms"])
s=d["settled"]; print("settled", s["value"], s["ms_per_step"], s["steps_timed"], s["timed_seconds"], s["breakdown_ms"], s["neighbourhood"])
print(d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
P
( time timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2>> $OUT/bench_default.err ) 2>&1 | grep real
