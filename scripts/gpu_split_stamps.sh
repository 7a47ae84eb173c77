#!/bin/bash
set -u
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
OUT=gpurun_out/split_stamps.log
: > $OUT
for k in ${KERNELS:-lite lite_split}; do
  FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_$k timeout -k 10 120 python benchmarks/a64_stamps.py c3 >> $OUT 2>&1 || exit 4
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    print(j['kernel'][16:], j['ms'], j['tflops'], 'low', j.get('seam_steps_low'), 'high', j.get('seam_steps_high'), 'st2 A|B', j.get('seam_step2_AB'), 'st3 A|B', j.get('seam_step3_AB'), 'seam', j['seam_steps_cyc'], 'epi', j['epilogue_cyc_median'], 'clk', j['clock_ghz'], 'kern', j['kernel_cyc_median'], 'all', j.get('all_jobs_cyc'))
"
