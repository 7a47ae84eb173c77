#!/bin/bash
# issue-cost microbenchmarks of asm/microbench.py (filler patterns between MFMAs, and the mb_solo_* cases without MFMAs):
# build here (no GPU needed), run through gpurun:  bash scripts/gpu_microbench.sh build && gpurun -- 'bash scripts/gpu_microbench.sh run'
set -eu
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
if [ "${1:-build}" = build ]; then
  python -m flash_attention_dlrs_amd.csrc.asm.microbench scripts/probes/mb.s
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c -o scripts/probes/mb.o scripts/probes/mb.s
  $LLVM/ld.lld -shared -o scripts/probes/mb.hsaco scripts/probes/mb.o
  hipcc --offload-arch=gfx950 -O2 -o scripts/probes/mb_run scripts/probes/mb_run.hip
else
  timeout -k 10 120 scripts/probes/mb_run scripts/probes/mb.hsaco scripts/probes/mb.s.names | tee gpurun_out/microbench.txt
fi
