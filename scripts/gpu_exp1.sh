#!/bin/bash
# experiment batch: ablations + causal launch-order sweep
set -u
echo "== ablations (timing only)"
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_abl.so python benchmarks/variants.py --pairs c3_noncausal:mfma16p_w8,c3_noncausal:abl_noexp,c3_noncausal:abl_nosum,c3_noncausal:abl_nomax,c3_noncausal:abl_all 2>&1 | grep pair
for g in 1 2 4 8 16; do
  echo "== causal group $g"
  FA2_CAUSAL_GROUP=$g python benchmarks/variants.py --pairs c3:mfma16p_w8,c3:mfma16p --rounds 4 2>&1 | grep pair
done
