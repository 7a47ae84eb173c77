#!/bin/bash
set -u
python -m pytest tests/test_fwd_parity.py -m gpu -q --timeout=600 2>&1 | tail -3
echo "== ablations (timing only)"
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_abl.so python benchmarks/variants.py --pairs c3_noncausal:mfma16p_w8,c3_noncausal:abl_nobar,c3_noncausal:abl_noload,c3_noncausal:abl_skeleton,c3_noncausal:abl_all,c3_noncausal:mfma16p_w8_x2 2>&1 | grep pair
echo "== x2"
FA2_CAUSAL_GROUP=2 python benchmarks/variants.py --pairs c3_noncausal:mfma16p_w8,c3_noncausal:mfma16p_w8_x2,c3_noncausal:mfma16p_x2,c3:mfma16p_w8,c3:mfma16p_w8_x2,c3:mfma16p,c3:mfma16p_x2 2>&1 | grep pair
