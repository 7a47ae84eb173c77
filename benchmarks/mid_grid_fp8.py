#!/usr/bin/env python3
"""fp8 (e4m3) counterpart of mid_grid.py: mfma8x (8 waves) vs mfma8x_w4 vs mfma8u -- and, non-causal with N a multiple of 256, the
generated a8 -- over grids of 64..2048 256-row tiles.  MID_GRID_SPREAD: standard deviation of the inputs (default 1: bench.py's)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, it=20):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / it * 1e3)
    return best


for causal in ((False,) if os.environ.get("MID_GRID_NONCAUSAL") else (True,) if os.environ.get("MID_GRID_CAUSAL") else (False, True)):
    for N in [int(x) for x in os.environ.get("MID_GRID_N", "512,1024,2048,4096,8192").split(",")]:
        for BH in (8, 16, 24, 32, 48, 64, 128):
            wg256 = BH * ((N + 255) // 256)
            if wg256 < 64 or wg256 > 2100:
                continue
            Q, K, V = ((torch.randn(1, BH, N, 128, device=dev) * float(os.environ.get("MID_GRID_SPREAD", "1.0"))).to(torch.float8_e4m3fn) for _ in range(3))
            r = {"causal": causal, "N": N, "BH": BH, "wg256": wg256}
            cands = ("mfma8x", "mfma8x_w4") + (() if N < 256 else ("a8",))     # (mfma8u: experiments library only)
            for v in ("auto",) + cands:
                r[v] = round(t(lambda: flash_attention_forward(Q, K, V, dev, causal=causal, variant=v)), 1)
            r["best"] = min(cands, key=lambda k: r[k])
            print(json.dumps(r), flush=True)
