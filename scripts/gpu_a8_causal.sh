#!/bin/bash
# causal a8: parity, then A/B against fa2_mfma8x on causal fp8 shapes (N(0, 1) inputs)
set -u
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_fwd_parity.py -q -x -k "fp8 or a8" 2>&1 | tail -4 || exit 2
timeout -k 10 500 python benchmarks/variants.py --rounds 7 --iters 10 --fp8-spread 1.0 --pairs c3_fp8:mfma8x,c3_fp8:a8,fp8_8k_causal:mfma8x,fp8_8k_causal:a8,fp8_2k_causal:mfma8x,fp8_2k_causal:a8 2>&1 | grep pair || exit 3
