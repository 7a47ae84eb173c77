#!/bin/bash
# round 3, late: power probe with the hybrid MFMA mix, rocprofv3 of c5's per-GPU shard on the table's final choice, the vs-torch sweeps
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 240 scripts/probes/pw_run scripts/probes/pw.hsaco scripts/probes/pw.s.names 1.5 > gpurun_out/powerprobe_mix.txt 2>&1 || { tail -5 gpurun_out/powerprobe_mix.txt; exit 2; }
grep -E "pw_(32|16|mix)_(bare|full)" gpurun_out/powerprobe_mix.txt
PROF_OUT=prof_c5m BENCH_ARGS="--config c5_per_gpu --steps 6 --warmup 2 --no-cpu-baseline --no-extras" timeout -k 10 900 bash scripts/gpu_prof.sh > gpurun_out/prof_c5m.log 2>&1 || { tail -5 gpurun_out/prof_c5m.log; exit 3; }
find gpurun_out/prof_c5m -name "*.csv" -size +2M -delete
timeout -k 10 300 python benchmarks/vs_torch.py --dtype bf16 > gpurun_out/vs_torch_bf16_r03.jsonl 2> gpurun_out/vs_torch_bf16_r03.err || { tail -5 gpurun_out/vs_torch_bf16_r03.err; exit 4; }
tail -3 gpurun_out/vs_torch_bf16_r03.jsonl
