#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box via gpurun).  Kernel trace + stats first, then the
# PMC counters in their own passes (never combined with trace domains other than the kernel trace).
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PROF_OUT:-prof}
rm -rf $OUT && mkdir -p $OUT
ARGS="${BENCH_ARGS:---steps 20 --warmup 5 --no-cpu-baseline --no-extras}"
echo "=== stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 2; }
tail -2 $OUT/stats.log
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  echo "=== pmc $grp"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py $ARGS > $OUT/pmc_$tag.log 2>&1 || { tail -5 $OUT/pmc_$tag.log; exit 3; }
done
find $OUT -name "*.csv" | head -40
