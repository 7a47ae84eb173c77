#!/bin/bash
# a8: K / V rings of two buffers (vmcnt(0) at the step's barrier: a DMA piece has ~1 300 cycles = 0.6 us to land) against four (vmcnt(4))
set -u
cd "$(dirname "$0")/.."
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 6 --fp8-spread 1.0 --pairs c5_per_gpu:a8,c5_per_gpu:a8:FA2_A64_KERNEL=fa2_fwd_a8_e4m3_n_ring4,c5_per_gpu:a8:FA2_A64_KERNEL=fa2_fwd_a8_e4m3_n_ring2,fp8_4k:a8,fp8_4k:a8:FA2_A64_KERNEL=fa2_fwd_a8_e4m3_n_ring4 2>&1 | grep pair || exit 3
