#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bwd
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/benchmarks/bench_bwd.py --configs ${CFG:-c3} > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 2; }
tail -3 $OUT/stats.log
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
head -8 $f | cut -c1-220
