#!/bin/bash
set -u
PAIRS="c3:mfma16d,c3:mfma16d:FA2_WG_PER_SLOT=1000,c3:mfma16d:FA2_WG_PER_SLOT=2,c3_noncausal:mfma16d,c3_noncausal:mfma16d:FA2_WG_PER_SLOT=1000,c4_per_gpu:mfma16d,c4_per_gpu:mfma16d:FA2_WG_PER_SLOT=1000"
for r in 1 2; do
  echo "== base"; FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_base.so timeout -k 10 200 python benchmarks/variants.py --pairs c3:mfma16d,c3_noncausal:mfma16d,c4_per_gpu:mfma16d --rounds 5 2>&1 | grep pair
  echo "== new"; timeout -k 10 200 python benchmarks/variants.py --pairs $PAIRS --rounds 5 2>&1 | grep pair
done | tee gpurun_out/exp5_ab.log
