#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -n 6
timeout 100 python bench.py --steps 100 --warmup 10 --cpu-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms'])"
