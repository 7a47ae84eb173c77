#!/bin/bash
# ragged a16: parity, then A/B against a64's ragged form on long ragged shapes
set -u
cd "$(dirname "$0")/.."
timeout -k 10 800 python -m pytest tests/test_a64_parity.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -3 || exit 2
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 10 --pairs ragged_8k:a64,ragged_8k:a16,ragged_4k:a64,ragged_4k:a16,ragged_8k_causal:a64,ragged_8k_causal:a16 2>&1 | grep pair
