#!/bin/bash
# job-level stamps of the a64 and a16 kernels (non-causal) and of a16's timing-only ablations
set -u
cd "$(dirname "$0")/.."
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
OUT=gpurun_out/a16_stamps.log
: > $OUT
for k in fa2_fwd_a64_bf16_n_lite fa2_fwd_a16_bf16_n_lite ${ABLS:-fa2_fwd_a16_bf16_n_mfmaonly fa2_fwd_a64_bf16_n_mfmaonly fa2_fwd_a16_bf16_n_nostart fa2_fwd_a16_bf16_n_nofinish fa2_fwd_a16_bf16_n_nolds fa2_fwd_a16_bf16_n_nobarrier fa2_fwd_a16_bf16_n_nofecv fa2_fwd_a16_bf16_n_nomx}; do
  FA2_A64_KERNEL=$k timeout -k 10 120 python benchmarks/a64_stamps.py ${CFG:-c3_noncausal} >> $OUT 2>&1 || exit 3
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    print(j['config'], j['kernel'][8:], j['ms'], j['tflops'], 'step', j.get('cyc_per_step_loop'), 'seam', j['seam_steps_cyc'], 'epi', j['epilogue_cyc_median'], 'clk', j['clock_ghz'], 'kern', j['kernel_cyc_median'], 'all', j.get('all_jobs_cyc'))
"
