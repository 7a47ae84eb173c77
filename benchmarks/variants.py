#!/usr/bin/env python3
"""A/B harness: time several (config, kernel variant) pairs interleaved in ONE process
(cdna_hip_programming.md rule 24) and print TFLOP/s per pair (median and best of the rounds).

    python benchmarks/variants.py --pairs c3:mfma16_w8,c3:mfma16,c3_noncausal:mfma16_w8 --rounds 5 --iters 10
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, PEAK_TFLOPS, TORCH_DTYPE, flops  # noqa: E402
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="c3:mfma16_w8,c3:mfma16,c3_noncausal:mfma16_w8,c3_noncausal:mfma16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--fp8-spread", type=float, default=0.5, help="standard deviation of the fp8 inputs (bench.py draws N(0, 1))")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    # pair = config:variant[:ENV=VALUE[+ENV=VALUE...]] -- the env settings are applied around that pair's launches
    # (the library reads its tuning knobs per call)
    triples = [(p.split(":") + [""])[:3] for p in args.pairs.split(",")]
    envs = {f"{c}:{v}" + (f":{e}" if e else ""): dict(kv.split("=") for kv in e.split("+")) if e else {} for c, v, e in triples}
    pairs = [(c, v + (f":{e}" if e else "")) for c, v, e in triples]

    def launch(cfg, var, Q, K, V):
        env = envs[f"{cfg}:{var}"]
        for k, val in env.items():
            os.environ[k] = val
        flash_attention_forward(Q, K, V, dev, causal=CONFIGS[cfg]["causal"], variant=var.split(":")[0], scale=float(env.get("SCALE", 1.0)))
        for k in env:
            os.environ.pop(k, None)
    data = {}
    for cfg, _ in pairs:
        if cfg not in data:
            c = CONFIGS[cfg]
            torch.manual_seed(42)
            data[cfg] = tuple((torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev) *
                               (args.fp8_spread if c["dtype"] == "fp8" else 1.0)).to(TORCH_DTYPE[c["dtype"]]) for _ in range(3))
    res = {f"{c}:{v}": [] for c, v in pairs}
    for cfg, var in pairs:  # warm-up
        Q, K, V = data[cfg]
        for _ in range(3):
            launch(cfg, var, Q, K, V)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for cfg, var in pairs:
            Q, K, V = data[cfg]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                launch(cfg, var, Q, K, V)
            b.record()
            torch.cuda.synchronize()
            res[f"{cfg}:{var}"].append(a.elapsed_time(b) / args.iters)
    for k, ms in res.items():
        c = CONFIGS[k.split(":")[0]]
        ms = sorted(ms)
        med, best = ms[len(ms) // 2], ms[0]
        tf = lambda t: flops(c) / (t * 1e-3) / 1e12
        print(json.dumps({"pair": k, "ms_median": round(med, 4), "tflops_median": round(tf(med), 1),
                          "tflops_best": round(tf(best), 1),
                          "pct_peak_median": round(100 * tf(med) / PEAK_TFLOPS[c["dtype"]], 1)}))


if __name__ == "__main__":
    main()
