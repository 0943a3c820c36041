#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q 2>&1 | tail -n 4
timeout 100 python bench.py --steps 100 --warmup 10 --cpu-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms'])"
timeout 100 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --settle 2000 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms'])"
