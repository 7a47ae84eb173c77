#!/bin/bash
# split row map of the causal a64 kernel: bit-equality with the base kernel, then A/B in one process
set -u
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so
timeout -k 10 300 python benchmarks/a64_variant_equal.py base split > gpurun_out/split_equal.log 2>&1 || { tail -30 gpurun_out/split_equal.log; exit 2; }
grep -v amdgpu.ids gpurun_out/split_equal.log
P="c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_base,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_split,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_split_late"
timeout -k 10 300 python benchmarks/variants.py --pairs $P --rounds 9 --iters 20 > gpurun_out/split_ab.log 2>&1 || { tail -5 gpurun_out/split_ab.log; exit 3; }
grep pair gpurun_out/split_ab.log
