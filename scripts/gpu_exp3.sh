#!/bin/bash
# mfma16s (16x16x32 shape) parity + A/B against mfma16d; static priority and -fno-honor-nans A/B
set -u
mkdir -p gpurun_out
echo "=== parity (mfma16s)"
timeout -k 10 600 python -m pytest tests/test_fwd_parity.py -m gpu -q -x --timeout=500 -k "seeded or golden or full_size or rescale or canary or ragged" > gpurun_out/exp3_pytest.log 2>&1
rc=$?; tail -n 15 gpurun_out/exp3_pytest.log; echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
echo "=== A/B default lib"
timeout -k 10 300 python benchmarks/variants.py --rounds 7 --pairs c3:mfma16d,c3:mfma16s,c3:mfma16d:FA2_FLAGS=1,c3:mfma16s:FA2_FLAGS=1,c3_noncausal:mfma16d,c3_noncausal:mfma16s,c3_noncausal:mfma16d:FA2_FLAGS=1,c3_noncausal:mfma16s:FA2_FLAGS=1 2>&1 | grep pair | tee gpurun_out/exp3_ab.log
echo "=== A/B nn lib"
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_nn.so timeout -k 10 300 python benchmarks/variants.py --rounds 7 --pairs c3:mfma16d,c3:mfma16s,c3_noncausal:mfma16d,c3_noncausal:mfma16s 2>&1 | grep pair | tee gpurun_out/exp3_ab_nn.log
