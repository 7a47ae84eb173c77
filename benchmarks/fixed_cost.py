#!/usr/bin/env python3
"""Per-workgroup fixed cost (prologue + epilogue + dispatch) of the default bf16 kernel: one full round of 256
workgroups (one per CU) at N = 2048 / 4096 / 8192 keys -- same workgroup count, 32 / 64 / 128 loop iterations."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward
dev = torch.device("cuda:0")
res = {}
for N, BH in ((2048, 32), (4096, 16), (8192, 8), (16384, 4)):
    torch.manual_seed(0)
    Q, K, V = (torch.randn(1, BH, N, 128, device=dev).bfloat16() for _ in range(3))
    for _ in range(20):
        flash_attention_forward(Q, K, V, dev, variant="mfma16d")
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            flash_attention_forward(Q, K, V, dev, variant="mfma16d")
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 50 * 1e3)
    res[N] = sorted(ts)[2]
    print(json.dumps({"N": N, "BH": BH, "workgroups": BH * N // 256, "iterations": N // 64, "us": round(res[N], 2)}))
c = (res[8192] - res[4096]) / 64
print(json.dumps({"us_per_iteration": round(c, 3), "fixed_us_N4096": round(res[4096] - 64 * c, 2),
                  "fixed_us_N2048": round(res[2048] - 32 * c, 2), "fixed_us_N16384": round(res[16384] - 256 * c, 2)}))
