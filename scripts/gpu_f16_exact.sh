#!/bin/bash
# f16 causal: the firing path of a lazily masked diagonal tile leaves again when the exact maximum does not pass the threshold -- A/B
# against the path without that exit (experiments library), then the f16 / bf16 parity tests of the generated kernels
set -u
cd "$(dirname "$0")/.."
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so timeout -k 10 400 python benchmarks/variants.py --rounds 9 --iters 20 --pairs c3_fp16:a64,c3_fp16:a64:FA2_A64_KERNEL=fa2_fwd_a64_f16_c_noexact,causal_2k_fp16:a64,causal_2k_fp16:a64:FA2_A64_KERNEL=fa2_fwd_a64_f16_c_noexact 2>&1 | grep pair || exit 2
timeout -k 10 600 python -m pytest tests/test_a64_parity.py tests/test_fuzz_gpu.py -q -x 2>&1 | tail -3
