#!/bin/bash
# Round-end evidence run: smoke, the whole GPU test suite, the default bench line, rocprofv3 passes (forward c3),
# backward kernel stats, the reference-style sweeps.  A step that is killed stops the chain.
set -u
mkdir -p gpurun_out
step() {
    local name=$1 to=$2; shift 2
    echo "=== $name"
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    tail -n ${TAILN:-6} "gpurun_out/$name.log"
    echo "=== $name rc=$rc"
    if [ $rc -gt 1 ]; then echo "step $name was killed or crashed (rc=$rc): stopping"; exit $rc; fi
    return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step pytest_gpu 1100 python -m pytest tests -m gpu -q --timeout=900 --maxfail=8
step bench 400 python bench.py --steps 100 --warmup 20
rm -rf gpurun_out/prof
step prof 900 bash scripts/gpu_prof.sh
rm -rf gpurun_out/prof_bwd
CFG=c3 step prof_bwd 300 bash scripts/gpu_bwd_prof.sh
step bench_bwd 300 python benchmarks/bench_bwd.py --torch
