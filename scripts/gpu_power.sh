#!/bin/bash
# power / clock probe of the two bf16 MFMA shapes under the a64 kernel's filler load (asm/powerprobe.py):
#   bash scripts/gpu_power.sh build   (no GPU needed)   &&   gpurun -- 'bash scripts/gpu_power.sh run'
set -eu
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
if [ "${1:-build}" = build ]; then
  python -m flash_attention_dlrs_amd.csrc.asm.powerprobe scripts/probes/pw.s
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c -o scripts/probes/pw.o scripts/probes/pw.s
  $LLVM/ld.lld -shared -o scripts/probes/pw.hsaco scripts/probes/pw.o
  hipcc --offload-arch=gfx950 -O2 -o scripts/probes/pw_run scripts/probes/pw_run.hip
else
  timeout -k 10 180 scripts/probes/pw_run scripts/probes/pw.hsaco scripts/probes/pw.s.names ${2:-1.5} | tee gpurun_out/powerprobe.txt
fi
