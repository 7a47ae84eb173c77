#!/usr/bin/env python3
"""A/B of several BUILDS of the library in one process (interleaved rounds): each libfa2_*.so given on the command
line is loaded with ctypes and timed on the same tensors.  Used for compiler-flag sweeps of a single kernel file.

    python benchmarks/lottery.py c3 mfma16d flash_attention_dlrs_amd/libfa2_hip.so flash_attention_dlrs_amd/libfa2_lot*.so
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE, flops  # noqa: E402
from flash_attention_dlrs_amd import _lib  # noqa: E402

cfg, variants, paths = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]  # every (build, variant) pair is timed
c = CONFIGS[cfg]
dev = torch.device("cuda:0")
torch.manual_seed(42)
Q, K, V = ((torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev) * (0.5 if c["dtype"] == "fp8" else 1.0)).to(TORCH_DTYPE[c["dtype"]]) for _ in range(3))
O = torch.empty_like(Q)
L = torch.empty(c["B"], c["H"], c["N"], 1, dtype=Q.dtype, device=dev)
i64p, vp = ctypes.POINTER(ctypes.c_int64), ctypes.c_void_p
arr = lambda v: (ctypes.c_int64 * len(v))(*v)
st = arr(Q.stride())
libs = []
for p in paths:
    l = ctypes.CDLL(os.path.abspath(p))
    l.fa2_fwd_variant.restype = ctypes.c_int
    l.fa2_fwd_variant.argtypes = [vp] * 5 + [i64p] * 5 + [ctypes.c_int32] * 6 + [ctypes.c_float, vp, ctypes.c_int32]
    for v in variants:
        libs.append((os.path.basename(p) + ":" + v, (l, v)))
dt = {"bf16": 2, "fp16": 1, "fp8": 4}[c["dtype"]]
stream = torch.cuda.current_stream(dev).cuda_stream


def run(lv):
    l, variant = lv
    rc = l.fa2_fwd_variant(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(), st, st, st, st,
                           arr((L.stride(0), L.stride(1))), c["B"], c["H"], c["N"], c["d"], dt, int(c["causal"]), 1.0,
                           stream, _lib.VARIANTS[variant])
    assert rc == 0, rc


import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _, l in libs:
        run(l)
    torch.cuda.synchronize()
res = {n: [] for n, _ in libs}
for r in range(9):
    order = libs if r % 2 == 0 else libs[::-1]
    for n, l in order:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run(l)
        b.record()
        torch.cuda.synchronize()
        res[n].append(a.elapsed_time(b) / 10)
for n, ms in res.items():
    ms = sorted(ms)
    print(json.dumps({"lib": n, "config": cfg, "tflops_median": round(flops(c) / (ms[len(ms) // 2] * 1e-3) / 1e12, 1),
                      "tflops_best": round(flops(c) / (ms[0] * 1e-3) / 1e12, 1)}))
