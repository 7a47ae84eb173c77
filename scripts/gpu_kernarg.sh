#!/bin/bash
# does HIP_FORCE_DEV_KERNARG change the kernel's pipeline fill (kernel-argument fetch) and the launch rate?
set -u
mkdir -p gpurun_out
OUT=gpurun_out/kernarg.log
: > $OUT
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v" >> $OUT
  HIP_FORCE_DEV_KERNARG=$v FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_lite timeout -k 10 120 python benchmarks/a64_stamps.py c3 >> $OUT 2>&1 || exit 4
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $OUT 2>&1 || exit 5
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    if 'fill_cyc' in j: print('stamps: ms', j['ms'], 'fill', j['fill_cyc'], 'kern cyc', j['kernel_cyc_median'])
    else: print('bench: value', j['value'], 'ms_per_step', j['ms_per_step'], 'kernel avg', j['roofline']['kernel_ms_avg'], 'median', j['roofline']['kernel_ms_median'])
"
