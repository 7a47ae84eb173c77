#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fwd_parity.py -m gpu -q -x --timeout=800 -k "${K:-seeded or ragged or rescale}" > gpurun_out/exp7_pytest.log 2>&1
rc=$?; tail -n 5 gpurun_out/exp7_pytest.log; echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
PAIRS=${PAIRS:-c3:mfma16d,c3:mfma16h,c3_noncausal:mfma16d,c3_noncausal:mfma16h,c4_per_gpu:mfma16d,c4_per_gpu:mfma16h}
timeout -k 10 300 python benchmarks/variants.py --pairs $PAIRS --rounds 7 2>&1 | grep pair | tee gpurun_out/exp7_ab.log
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so timeout -k 10 200 python benchmarks/stamps.py c3_noncausal 2>&1 | tail -8 | tee gpurun_out/stamps_nc.log
