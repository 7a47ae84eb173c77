#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bwd_parity.py -m gpu -q -x --timeout=800 > gpurun_out/bwd3_pytest.log 2>&1
rc=$?; tail -n 6 gpurun_out/bwd3_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python benchmarks/bench_bwd.py --configs c3,c3_noncausal,ref_bench 2>&1 | grep config | tee gpurun_out/bwd3_bench.log
