#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bwd_parity.py -m gpu -q -x --timeout=500 -k "larger or golden" > gpurun_out/bwd2_pytest.log 2>&1
rc=$?; tail -n 4 gpurun_out/bwd2_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python benchmarks/bench_bwd.py --torch 2>&1 | grep config | tee gpurun_out/bwd2_bench.log
