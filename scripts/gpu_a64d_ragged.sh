#!/bin/bash
# ragged a64d: parity (incl. the fuzz shapes), then A/B against the 8-wave kernels on ragged d = 64 shapes
set -u
cd "$(dirname "$0")/.."
timeout -k 10 800 python -m pytest tests/test_a64_parity.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -3 || exit 2
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 10 --pairs d64_ragged:a64d,d64_ragged:mfma16h,d64_ragged:mfma16d_w4,d64_ragged_causal:a64d,d64_ragged_causal:mfma16h 2>&1 | grep pair
