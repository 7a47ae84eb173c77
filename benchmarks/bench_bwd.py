#!/usr/bin/env python3
"""Backward timing at the BASELINE.json shapes: the three launches of include/fa2_bwd.h (D, dQ, dK/dV) through
flash_attention_backward, against torch SDPA's backward on the same device.  TFLOP/s with the 2.5 x forward
convention (10*B*H*N^2*d, halved for causal); the two-kernel scheme executes 14/10 of that (S and dP are
recomputed by both owners)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE  # noqa: E402
from flash_attention_dlrs_amd import flash_attention_backward, flash_attention_forward  # noqa: E402


def timeit(fn, iters=10, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c3,c3_noncausal,c2,ref_bench")
    ap.add_argument("--variants", default="auto")
    ap.add_argument("--torch", action="store_true", help="also time torch SDPA backward")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for cfg in args.configs.split(","):
        c = CONFIGS[cfg]
        torch.manual_seed(42)
        Q, K, V, dO = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev).to(TORCH_DTYPE[c["dtype"]]) for _ in range(4))
        O, L = flash_attention_forward(Q, K, V, dev, causal=c["causal"])
        fl = 10.0 * c["B"] * c["H"] * c["N"] ** 2 * c["d"] * (0.5 if c["causal"] else 1.0)
        for var in args.variants.split(","):
            ms = timeit(lambda: flash_attention_backward(Q, K, V, O, dO, L, dev, causal=c["causal"], variant=var))
            print(json.dumps({"config": cfg, "provider": f"hip:{var}", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)
        if args.torch:
            q, k, v = (t.detach().requires_grad_(True) for t in (Q, K, V))
            o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0, is_causal=c["causal"])
            ms = timeit(lambda: torch.autograd.grad(o, (q, k, v), dO, retain_graph=True))
            print(json.dumps({"config": cfg, "provider": "torch-sdpa", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}), flush=True)


if __name__ == "__main__":
    main()
