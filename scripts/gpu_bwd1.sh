#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bwd_parity.py tests/test_correctness.py -m gpu -q -x --timeout=800 > gpurun_out/bwd1_pytest.log 2>&1
rc=$?; tail -n 30 gpurun_out/bwd1_pytest.log; echo "pytest rc=$rc"
