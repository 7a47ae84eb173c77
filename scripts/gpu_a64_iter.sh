#!/bin/bash
# one development iteration of the generated assembly kernel on the GPU box: parity, A/B against mfma16h on the same device,
# then the job-level stamps of the timing-only ablations (diagnostic library)
set -u
timeout -k 10 600 python -m pytest tests/test_a64_parity.py -x -q -m gpu > gpurun_out/a64_pytest.log 2>&1 || { tail -30 gpurun_out/a64_pytest.log; exit 2; }
tail -1 gpurun_out/a64_pytest.log
timeout -k 10 300 python benchmarks/variants.py --pairs ${PAIRS:-c3:a64,c3:mfma16h,c3_noncausal:a64,c3_noncausal:mfma16h} --rounds 7 --iters 20 > gpurun_out/a64_bench.log 2>&1 || { tail -5 gpurun_out/a64_bench.log; exit 3; }
grep pair gpurun_out/a64_bench.log
[ -n "${ABLS:-}" ] || exit 0
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
OUT=gpurun_out/a64_stamps.log
: > $OUT
FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_lite timeout -k 10 120 python benchmarks/a64_stamps.py c3 >> $OUT 2>&1 || exit 4
for k in $ABLS; do
  FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_$k timeout -k 10 120 python benchmarks/a64_stamps.py c3_noncausal >> $OUT 2>&1 || exit 5
done
grep -v amdgpu.ids $OUT | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.strip()); continue
    j = json.loads(l)
    print(j['config'], j['kernel'][16:] or 'base', j['ms'], j['tflops'], 'step', j.get('cyc_per_step_loop'), 'seam', j['seam_steps_cyc'], 'epi', j['epilogue_cyc_median'], 'clk', j['clock_ghz'], 'kern', j['kernel_cyc_median'], 'job', j['job_cyc'])
"
