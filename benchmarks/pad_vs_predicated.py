#!/usr/bin/env python3
"""head sizes the matrix-core kernels do not take natively.  Times, per shape, (a) the library as it routes
the unpadded tensors, (b) a host pad to the next multiple of 8 / 4 (the d-predicated MFMA path), (c) a host pad to 64 / 128 (the
fast kernels) -- pad copies and the slice of O included in (b) and (c).

    python benchmarks/pad_vs_predicated.py
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward  # noqa: E402
from flash_attention_dlrs_amd.flash_attention_torch import pad_last_dim  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    for dtype in (torch.bfloat16, torch.float32):
        q = 8 if dtype != torch.float32 else 4
        for (B, H, N) in ((4, 32, 4096), (2, 8, 1024)):
            for d in ((100, 36, 20, 80, 96, 120) if dtype != torch.float32 else (50, 30, 96)):
                for causal in (False, True):
                    torch.manual_seed(1)
                    Q, K, V = (torch.randn(B, H, N, d, device=dev).to(dtype) for _ in range(3))
                    F = 4.0 * B * H * N * N * d * (0.5 if causal else 1.0)
                    row = {"dtype": str(dtype).replace("torch.", ""), "B": B, "H": H, "N": N, "d": d, "causal": causal}
                    row["as_is_ms"] = round(timed(lambda: flash_attention_forward(Q, K, V, dev, causal=causal)), 4)
                    for name, dp in (("pad_mult", (d + q - 1) // q * q), ("pad_pow2", 64 if d <= 64 else 128)):
                        if dp == d:
                            continue

                        def run():
                            O, L = flash_attention_forward(pad_last_dim(Q, dp), pad_last_dim(K, dp), pad_last_dim(V, dp), dev, causal=causal)
                            return O[..., :d]
                        row[name + "_ms"] = round(timed(run), 4)
                        row[name + "_d"] = dp
                    row["tflops_as_is"] = round(F / row["as_is_ms"] / 1e9, 1)
                    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
