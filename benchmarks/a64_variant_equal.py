#!/usr/bin/env python3
"""Two named variants of the causal a64 kernel (experiments library, FA2_A64_KERNEL) must give the same bits on the same inputs.

    FA2_HIP_LIB=flash_attention_dlrs_amd/libfa2_hip_exp.so python benchmarks/a64_variant_equal.py base split
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402

a, b = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
bad = 0
dt = torch.bfloat16
for (B, H, N) in ((1, 1, 256), (1, 3, 512), (2, 5, 768), (1, 8, 1024), (3, 8, 2048), (4, 32, 4096), (1, 2, 8192), (1, 7, 1280)):
    torch.manual_seed(N + H)
    Q, K, V = (torch.randn(B, H, N, 128, device=dev).to(dt) for _ in range(3))
    if N == 768:   # large scores on keys just ahead of their queries: the lazily masked diagonal and the exact firing path
        for q, ahead, gain in ((5, 3, 2.0), (40, 20, 4.0), (100, 60, 8.0), (300, 1, 6.0), (517, 50, 3.0), (600, 100, 5.0)):
            K[:, :, q + ahead] = (gain * Q[:, :, q].float()).to(dt)
    out = {}
    for v in (a, b):
        os.environ["FA2_A64_KERNEL"] = f"fa2_fwd_a64_bf16_c_{v}"
        O, L = flash_attention_forward(Q, K, V, dev, causal=True, variant="a64")
        torch.cuda.synchronize()
        out[v] = (O.clone(), L.clone())
    os.environ.pop("FA2_A64_KERNEL", None)
    eqO = torch.equal(out[a][0], out[b][0])
    eqL = torch.equal(out[a][1], out[b][1])
    ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), is_causal=True, scale=1.0)
    err = (out[b][0].float() - ref).abs().max().item()
    nd = (out[a][0] != out[b][0]).sum().item()
    print(f"bf16 B{B} H{H} N{N}: O equal {eqO} ({nd} elements differ), L equal {eqL}, |O_{b} - fp32 SDPA| max {err:.4f}", flush=True)
    bad += (not eqO) + (not eqL)
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
