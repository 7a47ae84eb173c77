#!/bin/bash
# f16: what do the rescales of the deferred maximum cost?  The product threshold (15.875: P < 65 504) against one that never fires
# (timing only: P overflows) and against bf16 on the same shape -- the reference bench's shape B8 H16 N4096 d128
set -u
cd "$(dirname "$0")/.."
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 10 --pairs ref_bench:a64,ref_bench:a64:FA2_A64_THR=60,ref_bench:a64:FA2_A64_THR=12,ref_bench_bf16:a64,ref_bench:a16,ref_bench:a16:FA2_A64_THR=60 2>&1 | grep pair
