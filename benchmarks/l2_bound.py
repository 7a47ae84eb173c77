#!/usr/bin/env python3
"""Upper bound of what K/V locality can buy: the same launch with every (b, h) reading ONE K/V slice (strides 0 over batch and
head: 2 MiB of K/V in all, L2-resident) against the real tensors, interleaved in one process.  Timing only -- the aliased run
computes a different (valid) problem with the same instruction stream.

    python benchmarks/l2_bound.py [--config c3] [--variant auto] [--rounds 7] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE, flops  # noqa: E402
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    c = CONFIGS[args.config]
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    Q, K, V = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev).to(TORCH_DTYPE[c["dtype"]]) for _ in range(3))
    Ka, Va = (t[:1, :1].expand(c["B"], c["H"], c["N"], c["d"]) for t in (K, V))
    Qa = Q[:1, :1].expand(c["B"], c["H"], c["N"], c["d"])
    arms = {"real": (Q, K, V), "kv_aliased": (Q, Ka, Va), "qkv_aliased": (Qa, Ka, Va)}
    res = {k: [] for k in arms}
    for q, k, v in arms.values():
        for _ in range(5):
            flash_attention_forward(q, k, v, dev, causal=c["causal"], variant=args.variant)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, (q, k, v) in arms.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                flash_attention_forward(q, k, v, dev, causal=c["causal"], variant=args.variant)
            b.record()
            torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) / args.iters)
    for name, ms in res.items():
        ms = sorted(ms)
        med = ms[len(ms) // 2]
        print(json.dumps({"config": args.config, "arm": name, "ms_median": round(med, 4),
                          "tflops_median": round(flops(c) / (med * 1e-3) / 1e12, 1), "tflops_best": round(flops(c) / (ms[0] * 1e-3) / 1e12, 1)}))


if __name__ == "__main__":
    main()
