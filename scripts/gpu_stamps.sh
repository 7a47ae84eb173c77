#!/bin/bash
# in-kernel stamps of the generated assembly kernel and its timing-only ablations (diagnostic library: make stamps)
set -u
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
OUT=gpurun_out/a64_stamps.log
: > $OUT
[ -n "${SKIPBASE:-}" ] || timeout -k 10 120 python benchmarks/a64_stamps.py c3 >> $OUT 2>&1 || exit 2
timeout -k 10 120 python benchmarks/a64_stamps.py c3_noncausal >> $OUT 2>&1 || exit 2
for k in ${ABLS:-mfmaonly nostart nofinish novread nokread nodma novmwait nobarrier}; do
  FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_$k timeout -k 10 120 python benchmarks/a64_stamps.py c3_noncausal >> $OUT 2>&1 || exit 3
done
cat $OUT
