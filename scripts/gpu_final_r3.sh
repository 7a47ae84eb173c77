#!/bin/bash
# Round-3 evidence run on the final binary: smoke, the whole GPU suite, bench lines (the driver's protocol and a longer one), rocprofv3
# passes of c3 (a64), c4's per-GPU shard (a16) and c5's (a8), backward stats, A/B lines.  A step that is killed stops the chain.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
step() {
    local name=$1 to=$2; shift 2
    echo "=== $name"
    timeout -k 10 "$to" "$@" > "gpurun_out/final/$name.log" 2>&1
    local rc=$?
    tail -n ${TAILN:-4} "gpurun_out/final/$name.log"
    echo "=== $name rc=$rc"
    if [ $rc -gt 1 ]; then echo "step $name was killed or crashed (rc=$rc): stopping"; exit $rc; fi
    return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
[ -n "${SKIP_TESTS:-}" ] || step pytest_gpu 1100 python -m pytest tests -m gpu -q --timeout=900 --maxfail=8
step bench_20_5 400 python bench.py --steps 20 --warmup 5
step bench_100_20 400 python bench.py --steps 100 --warmup 20 --no-cpu-baseline
step bench_c3nc 300 python bench.py --config c3_noncausal --steps 50 --warmup 10 --no-cpu-baseline --no-extras
step bench_c4 300 python bench.py --config c4_per_gpu --steps 30 --warmup 10 --no-cpu-baseline --no-extras
step bench_c5 300 python bench.py --config c5_per_gpu --steps 10 --warmup 3 --no-cpu-baseline --no-extras
step bench_c2 300 python bench.py --config c2 --steps 200 --warmup 50 --no-cpu-baseline --no-extras
PROF_OUT=prof_c3 step prof_c3 900 bash scripts/gpu_prof.sh
PROF_OUT=prof_c4 BENCH_ARGS="--config c4_per_gpu --steps 10 --warmup 3 --no-cpu-baseline --no-extras" step prof_c4 900 bash scripts/gpu_prof.sh
PROF_OUT=prof_c5 BENCH_ARGS="--config c5_per_gpu --steps 6 --warmup 2 --no-cpu-baseline --no-extras" step prof_c5 900 bash scripts/gpu_prof.sh
find gpurun_out/prof_c3 gpurun_out/prof_c4 gpurun_out/prof_c5 -name "*.csv" -size +2M -delete
step variants 400 python benchmarks/variants.py --rounds 7 --iters 20 --pairs c3:a64,c3:a16,c3:mfma16h,c3_noncausal:a64,c3_noncausal:a16,c4_per_gpu:a64,c4_per_gpu:a16,causal_8k:a64,causal_8k:a16,c5_per_gpu:mfma8x,c5_per_gpu:a8
