#!/usr/bin/env python3
"""Is the kernel clock/power-limited?  Time the same launch on N(0,1) data, on d^-1/4-scaled data and on zeros
(zeros switch far fewer bits: a large speed-up on zeros at identical instruction counts = DVFS give-back,
MI355X_MICROARCH.md 'DVFS give-back')."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE, flops
from flash_attention_dlrs_amd import flash_attention_forward

dev = torch.device("cuda:0")
for cfg in sys.argv[1:] or ["c3_noncausal", "c3"]:
    c = CONFIGS[cfg]
    dt = TORCH_DTYPE[c["dtype"]]
    torch.manual_seed(42)
    shape = (c["B"], c["H"], c["N"], c["d"])
    data = {"randn": tuple(torch.randn(*shape, device=dev).to(dt) for _ in range(3))}
    data["scaled"] = tuple((x.float() * s).to(dt) for x, s in zip(data["randn"], (c["d"] ** -0.25, c["d"] ** -0.25, 1.0)))
    data["zeros"] = tuple(torch.zeros(*shape, device=dev, dtype=dt) for _ in range(3))
    res = {k: [] for k in data}
    for _ in range(5):
        for k, (Q, K, V) in data.items():
            for _ in range(5):
                flash_attention_forward(Q, K, V, dev, causal=c["causal"])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                flash_attention_forward(Q, K, V, dev, causal=c["causal"])
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 20)
    print(json.dumps({"config": cfg, **{k: round(flops(c) / (sorted(v)[len(v) // 2] * 1e-3) / 1e12, 1) for k, v in res.items()}}))
