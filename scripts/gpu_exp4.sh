#!/bin/bash
# persistent job loop + cross-job prefetch (mfma16d): parity, then A/B against the previous build (libfa2_hip_base.so)
set -u
mkdir -p gpurun_out
echo "=== parity"
timeout -k 10 900 python -m pytest tests/test_fwd_parity.py tests/test_correctness.py -m gpu -q -x --timeout=800 > gpurun_out/exp4_pytest.log 2>&1
rc=$?; tail -n 15 gpurun_out/exp4_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
PAIRS=${PAIRS:-c3:mfma16d,c3_noncausal:mfma16d,c3:mfma16d_w4,c4_per_gpu:mfma16d,ref_bench:mfma16d}
for r in 1 2; do
  for lib in libfa2_hip_base.so libfa2_hip.so; do
    echo "== $lib"
    FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/$lib timeout -k 10 200 python benchmarks/variants.py --pairs $PAIRS --rounds 5 2>&1 | grep pair
  done
done | tee gpurun_out/exp4_ab.log
