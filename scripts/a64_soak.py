#!/usr/bin/env python3
"""One-off soak of the generated assembly kernel: random (B, H, N, dtype, causal, scale, storage layout) against fp32 SDPA on the
device, many more cases than tests/test_fuzz_gpu.py runs.   python scripts/a64_soak.py [cases] [seed]"""
import math
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_attention_dlrs_amd as fa  # noqa: E402

DEV = torch.device("cuda:0")
TOL = {torch.bfloat16: 2.5e-2, torch.float16: 4e-3}
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = random.Random(seed)
bad = 0
for k in range(n):
    N = rng.choice([256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 4095, 4096]) if rng.random() < 0.3 else rng.randint(256, 4500)
    B, H = rng.choice([(1, 1), (1, 2), (1, 3), (2, 4), (1, 8), (3, 8), (2, 5), (4, 16), (1, 40), (2, 33), (8, 32)])
    if B * H * N > 600_000:
        B, H = 1, max(1, 600_000 // N // 4)
    dtype = rng.choice([torch.bfloat16, torch.float16])
    causal = rng.random() < 0.5
    scale = rng.choice([1.0, 1.0, 128 ** -0.5, 0.3, 2.0])
    layout = rng.choice(["contiguous", "bnhd", "padded_rows"])
    g = torch.Generator().manual_seed(seed * 100003 + k)
    amp = 0.7 if scale >= 1.0 else 2.0
    mk = {"contiguous": lambda: (torch.randn(B, H, N, 128, generator=g) * amp).to(dtype).to(DEV),
          "bnhd": lambda: (torch.randn(B, N, H, 128, generator=g) * amp).to(dtype).to(DEV).transpose(1, 2),
          "padded_rows": lambda: (torch.randn(B, H, N, 136, generator=g) * amp).to(dtype).to(DEV)[..., :128]}[layout]
    Q, K, V = mk(), mk(), mk()
    O, L = fa.flash_attention_forward(Q, K, V, DEV, causal=causal, scale=scale, variant="a64")
    ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=scale, is_causal=causal)
    err = (O.float() - ref).abs().max().item()
    ok = math.isfinite(err) and err <= TOL[dtype] * max(1.0, ref.abs().max().item()) and torch.isfinite(L.float()).all().item()
    if not ok:
        bad += 1
        print("FAIL", dict(B=B, H=H, N=N, dtype=str(dtype), causal=causal, scale=scale, layout=layout, err=err), flush=True)
    if k % 50 == 49:
        print(f"{k + 1} cases, {bad} failures", flush=True)
print("soak done:", n, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
