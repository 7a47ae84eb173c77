#!/usr/bin/env python3
"""Diagnostic: per-phase cycle sums of workgroup 0 from a -DFA2_STAMPS build of fa2_mfma16h.hip (default) or
fa2_mfma8x.hip (cdna_hip_programming.md section 7, in-kernel stamps).  Run with FA2_HIP_LIB=.../libfa2_hip_stamps.so.

    python benchmarks/stamps.py <config> [variant]      # variant: mfma16h (default) | mfma8x | mfma8x_w4"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE  # noqa: E402
from flash_attention_dlrs_amd import _lib, flash_attention_forward  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_noncausal"
variant = sys.argv[2] if len(sys.argv) > 2 else "mfma16h"
c = CONFIGS[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(42)
Q, K, V = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev).to(TORCH_DTYPE[c["dtype"]]) for _ in range(3))
for _ in range(20):
    flash_attention_forward(Q, K, V, dev, causal=c["causal"], variant=variant)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
rc = _lib.lib().fa2_debug_read_stamps(buf)
names = ["dma_issue", "A_qk", "A_pv", "A_decide", "B_qk", "B_pv", "B_decide", "dma_wait", "barrier"]
if variant.startswith("mfma8x"):
    names = ["body", "dma_wait", "barrier"]
for w in range(8):
    row = [buf[w * 16 + k] for k in range(16)]
    trips = max(row[15], 1)
    per = {n: round(row[k] / trips, 1) for k, n in enumerate(names)}
    per["total"] = round(sum(row[:len(names)]) / trips, 1)
    print(json.dumps({"wave": w, "trips": trips, **per}))
