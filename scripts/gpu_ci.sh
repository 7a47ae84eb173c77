#!/bin/bash
# Runs on the GPU box (via gpurun): smoke -> GPU parity tests -> bench.  A step that times out or is
# killed (rc > 1) stops the chain: no further GPU step is started after a hang.
set -u
mkdir -p gpurun_out
step() {  # step <name> <timeout_s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== $name"
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    tail -n 25 "gpurun_out/$name.log"
    echo "=== $name rc=$rc"
    if [ $rc -gt 1 ]; then echo "step $name was killed or crashed (rc=$rc): stopping"; exit $rc; fi
    return 0
}
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step pytest_gpu 1000 python -m pytest tests -m gpu -q --timeout=900 "${PYTEST_ARGS:---maxfail=8}"
step bench 400 python bench.py --steps 30 --warmup 5
