#!/bin/bash
# O row stores with the non-temporal / system-scope hints (the kernel never reads O back): A/B against plain stores, one process
set -u
cd "$(dirname "$0")/.."
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so timeout -k 10 500 python benchmarks/variants.py --rounds 9 --iters 20 --pairs c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_base,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_o_nt,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_o_sc1,c3_noncausal:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_base,c3_noncausal:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_o_nt 2>&1 | grep pair
