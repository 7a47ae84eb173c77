#!/bin/bash
# the first job's loads: Q rows and K(0) first with a counted wait, against everything-then-wait (experiments library, one process)
set -u
cd "$(dirname "$0")/.."
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so timeout -k 10 500 python benchmarks/variants.py --rounds 11 --iters 20 --pairs c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_base,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_prologue_old,c3_noncausal:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_base,c3_noncausal:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_prologue_old,causal_2k:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_base,causal_2k:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_prologue_old 2>&1 | grep pair
