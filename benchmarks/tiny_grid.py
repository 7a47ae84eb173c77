#!/usr/bin/env python3
"""GPU microseconds per forward launch on SMALL grids (fewer 128-row tiles than CUs), measured by replaying a HIP graph
of 20 launches (the eager path through Python + ctypes costs ~15 us per call and hides everything below that):
the 128-row kernel `mfma16d_w4` against the key-split shapes of `mfma16k`.

    python benchmarks/tiny_grid.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, it=200):
    for _ in range(20):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / it * 1e3)
    return best


SHAPES = [(2, 8, N, 64, False) for N in (128, 256, 512, 1024, 2048)] + [(2, 8, 1024, 64, True), (3, 8, 1024, 64, False),
          (1, 8, 4096, 64, False), (2, 8, 512, 128, False), (2, 8, 1024, 128, False), (2, 8, 1024, 128, True),
          (3, 8, 1024, 128, False), (1, 8, 2048, 128, False), (1, 8, 4096, 128, True), (2, 8, 2048, 128, False)]
for B, H, N, d, causal in SHAPES:
    Q, K, V = (torch.randn(B, H, N, d, device=dev).to(torch.float16) for _ in range(3))
    r = {"shape": [B, H, N, d], "causal": causal, "wg128": B * H * ((N + 127) // 128),
         "eager_auto": round(t(lambda: flash_attention_forward(Q, K, V, dev, causal=causal)), 2)}
    for v in ("auto", "mfma16d_w4", "mfma16k", "mfma16k_r2k2") + (("mfma16k_r2k4",) if d == 64 else ()):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                flash_attention_forward(Q, K, V, dev, causal=causal, variant=v)
        r[v] = round(t(lambda: g.replay(), 20) / 20, 2)
    print(json.dumps(r), flush=True)
