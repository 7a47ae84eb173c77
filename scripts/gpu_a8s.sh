#!/bin/bash
# the block-scaled a8: parity, then A/B on c5's per-GPU shard against fa2_mfma8x and the unscaled form (experiments library)
set -u
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_fwd_parity.py -q -x -k "fp8 or a8" 2>&1 | tail -5 || exit 2
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so
for sp in 1.0 0.5; do
echo "=== spread $sp"
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 6 --fp8-spread $sp --pairs c5_per_gpu:mfma8x,c5_per_gpu:a8,c5_per_gpu:a8:FA2_A64_KERNEL=fa2_fwd_a8_e4m3_n_unscaled,fp8_4k:mfma8x,fp8_4k:a8 2>&1 | grep pair || exit 3
done
