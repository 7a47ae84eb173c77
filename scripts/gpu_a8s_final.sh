#!/bin/bash
# a8 (block-scaled) as the fp8 table choice: the fp8 / fuzz / property tests, the c5 bench line, a rocprofv3 pass
set -u
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "fp8 or a8 or fuzz or c5 or abi or table" 2>&1 | tail -4 || exit 2
timeout -k 10 300 python bench.py --config c5_per_gpu --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/bench_c5_a8s.log 2>&1 || exit 3
tail -1 gpurun_out/bench_c5_a8s.log | cut -c1-400
PROF_OUT=prof_c5s BENCH_ARGS="--config c5_per_gpu --steps 6 --warmup 2 --no-cpu-baseline --no-extras" timeout -k 10 900 bash scripts/gpu_prof.sh > gpurun_out/prof_c5s.log 2>&1 || { tail -5 gpurun_out/prof_c5s.log; exit 4; }
find gpurun_out/prof_c5s -name "*.csv" -size +2M -delete
echo DONE
