#!/bin/bash
# causal unit order: heads per XCD group (2 since round 2) with the downward light jobs -- A/B in one process
set -u
cd "$(dirname "$0")/.."
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so timeout -k 10 500 python benchmarks/variants.py --rounds 9 --iters 20 --pairs c3:a64:FA2_A64_GROUP=2,c3:a64:FA2_A64_GROUP=1,c3:a64:FA2_A64_GROUP=4,c3:a64:FA2_A64_GROUP=8,c3:a64:FA2_A64_GROUP=16,causal_2k:a64:FA2_A64_GROUP=2,causal_2k:a64:FA2_A64_GROUP=1,causal_2k:a64:FA2_A64_GROUP=4 2>&1 | grep pair
