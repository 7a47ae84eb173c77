#!/bin/bash
# job-level stamps of the timing-only ablations of the steady loop (diagnostic library; non-causal c3 shape)
set -u
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
mkdir -p gpurun_out
OUT=gpurun_out/abl_stamps.log
: > $OUT
for k in ${ABLS:-lite mfmaonly nobarrier nodma nokread novread nostart nofinish nomx nodec nofire nof noe nocv nofecv nolds nobar_nolds valuonly}; do
  FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_$k timeout -k 10 120 python benchmarks/a64_stamps.py c3_noncausal >> $OUT 2>&1 || exit 5
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    print(j['kernel'][16:] or 'base', 'step', j.get('cyc_per_step_loop'), 'tflops', j['tflops'], 'clk', j['clock_ghz'], 'seam', j['seam_steps_cyc'][:4], 'epi', j['epilogue_cyc_median'])
"
