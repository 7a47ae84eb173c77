#!/usr/bin/env python3
"""Where does the static tile table switch between the 128-row kernel (mfma16d_w4), the persistent 256-row kernels
(mfma16h, and for d = 128 the generated assembly kernel a64) and the key-split kernel (mfma16k)?  Eager microseconds per launch (bf16) over grids of 128..1024 tiles.

    python benchmarks/mid_grid.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, it=30):
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


ONLY_D = [int(x) for x in os.environ.get("MID_GRID_D", "128,64").split(",")]
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("MID_GRID_DTYPE", "bf16")]      # (f16 rescales every few tiles at scale 1)
BHS = [int(x) for x in os.environ.get("MID_GRID_BH", "8,12,16,24,32,48,64").split(",")]
NS = [int(x) for x in os.environ.get("MID_GRID_N", "1024,2048,4096,8192").split(",")]
for d in ONLY_D:
    for causal in (False, True):
        for N in NS:
            for BH in BHS:
                wg256 = BH * ((N + 255) // 256)
                if wg256 < 32 or wg256 > 1100:
                    continue
                Q, K, V = (torch.randn(1, BH, N, d, device=dev).to(DT) for _ in range(3))
                r = {"d": d, "causal": causal, "N": N, "BH": BH, "wg256": wg256}
                cands = ("mfma16d_w4", "mfma16h", "mfma16k", "mfma16k_r2k2") + (("a64",) if d == 128 and N >= 256 else ()) + (("a16",) if d == 128 and N % 256 == 0 and N >= 2048 else ()) + \
                    (("a64d", "mfma16k_r2k4") if d == 64 and N % 256 == 0 else ())
                for v in ("auto",) + cands:
                    r[v] = round(t(lambda: flash_attention_forward(Q, K, V, dev, causal=causal, variant=v)), 1)
                r["best"] = min(cands, key=lambda k: r[k])
                print(json.dumps(r), flush=True)
