#!/bin/bash
# job-level stamps of the fp8 kernel a8 and its timing-only ablations
set -u
cd "$(dirname "$0")/.."
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so A64_STAMPS_VARIANT=a8
OUT=gpurun_out/a8_stamps.log
: > $OUT
for k in lite ${ABLS:-mfmaonly nostart nofinish nolds nofecv nomx nobarrier nodma}; do
  FA2_A64_KERNEL=fa2_fwd_a8_e4m3_n_$k timeout -k 10 120 python benchmarks/a64_stamps.py ${CFG:-fp8_4k} ${DATA:-small} >> $OUT 2>&1 || exit 3
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    print(j['config'], j['kernel'][8:], j['ms'], j['tflops'], 'step', j.get('cyc_per_step_loop'), 'seam', j['seam_steps_cyc'], 'epi', j['epilogue_cyc_median'], 'clk', j['clock_ghz'], 'kern', j['kernel_cyc_median'], 'all', j.get('all_jobs_cyc'))
"
