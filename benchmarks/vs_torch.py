#!/usr/bin/env python3
"""Forward TFLOP/s of the default kernel choice against torch SDPA (ROCm's own flash kernel) over a grid of
shapes -- where is the static tile table weak?  One JSON line per shape.

    python benchmarks/vs_torch.py [--dtype bf16] [--tokens 65536]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tokens", type=int, default=65536)
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    dev = torch.device("cuda:0")
    H = 16
    for d in (64, 128):
        for N in (256, 512, 1024, 2048, 4096, 8192, 16384):
            B = max(1, args.tokens // N // H * 1) or 1
            B = max(1, args.tokens // (N * H) * H // H)
            for causal in (False, True):
                torch.manual_seed(0)
                Q, K, V = (torch.randn(B, H, N, d, device=dev).to(dt) for _ in range(3))
                fl = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
                iters = max(3, int(2e-2 / (fl / 8e14)))
                iters = min(iters, 200)
                t_hip = timeit(lambda: flash_attention_forward(Q, K, V, dev, causal=causal), iters)
                t_sd = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1.0, is_causal=causal), iters)
                print(json.dumps({"B": B, "H": H, "N": N, "d": d, "causal": causal, "hip_tflops": round(fl / t_hip * 1e-9, 1),
                                  "sdpa_tflops": round(fl / t_sd * 1e-9, 1), "ratio": round(t_sd / t_hip, 2)}), flush=True)


if __name__ == "__main__":
    main()
