#!/bin/bash
# ragged a8: parity (fp8 tests, fuzz), then A/B against fa2_mfma8x on ragged fp8 shapes (N(0, 1) inputs)
set -u
cd "$(dirname "$0")/.."
timeout -k 10 800 python -m pytest tests/test_fwd_parity.py tests/test_fuzz_gpu.py -q -x -k "fp8 or a8" 2>&1 | tail -3 || exit 2
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 10 --fp8-spread 1.0 --pairs fp8_ragged:mfma8x,fp8_ragged:a8,fp8_ragged_causal:mfma8x,fp8_ragged_causal:a8 2>&1 | grep pair
