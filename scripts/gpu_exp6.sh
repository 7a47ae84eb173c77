#!/bin/bash
set -u
mkdir -p gpurun_out
echo "=== parity"
timeout -k 10 900 python -m pytest tests/test_fwd_parity.py -m gpu -q -x --timeout=800 -k "${K:-seeded or golden or ragged or rescale or out_of_bounds or scale_ext}" > gpurun_out/exp6_pytest.log 2>&1
rc=$?; tail -n 12 gpurun_out/exp6_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
PAIRS=${PAIRS:-c3:mfma16d,c3:mfma16h,c3_noncausal:mfma16d,c3_noncausal:mfma16h,c4_per_gpu:mfma16d,c4_per_gpu:mfma16h,ref_bench:mfma16d,ref_bench:mfma16h}
timeout -k 10 300 python benchmarks/variants.py --pairs $PAIRS --rounds 7 2>&1 | grep pair | tee gpurun_out/exp6_ab.log
