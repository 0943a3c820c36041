#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q 2>&1 | tail -n 12
