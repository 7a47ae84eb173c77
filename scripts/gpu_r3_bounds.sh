#!/bin/bash
# round 3, first GPU call: where is the headroom?  bench protocol (20/5 as the driver runs it, 100/20), timing-only bounds
# (row-sum MFMAs, K/V locality), MFMA-shape power probe.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3_bounds
mkdir -p $O
for k in 1 2; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20_5_$k.json 2> $O/bench_20_5_$k.err || exit 2; done
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > $O/bench_100_20.json 2> $O/err || exit 2
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3_bounds/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['extras'].get('same_shape_no_mask'))
P
FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so timeout -k 10 300 python benchmarks/variants.py --rounds 7 --iters 20 --pairs \
c3:a64,c3:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_c_norowsum,c3_noncausal:a64,c3_noncausal:a64:FA2_A64_KERNEL=fa2_fwd_a64_bf16_n_norowsum 2>&1 | grep pair | tee $O/variants.jsonl
timeout -k 10 200 python benchmarks/l2_bound.py --config c3 2>&1 | grep arm | tee $O/l2_bound.jsonl
timeout -k 10 200 python benchmarks/l2_bound.py --config c3_noncausal 2>&1 | grep arm | tee -a $O/l2_bound.jsonl
bash scripts/gpu_power.sh run 1.5
