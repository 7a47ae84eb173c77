#!/usr/bin/env python3
"""Diagnostic: per-phase cycle sums of one key-owner workgroup of the backward (a -DSTAMP build of fa2_bwd_mfma16.hip
linked as libfa2_hip_bst.so).  slots: first products + softmax part, second products, DMA wait,
barrier; per wave (0-3 dK, 4-7 dV)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, TORCH_DTYPE
from flash_attention_dlrs_amd import _lib, flash_attention_backward, flash_attention_forward
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3_noncausal"]
dev = torch.device("cuda:0")
torch.manual_seed(42)
Q, K, V, dO = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev).to(TORCH_DTYPE[c["dtype"]]) for _ in range(4))
O, L = flash_attention_forward(Q, K, V, dev, causal=c["causal"])
for _ in range(10):
    flash_attention_backward(Q, K, V, O, dO, L, dev, causal=c["causal"])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
_lib.lib().fa2_debug_read_bwd_stamps(buf)
for w in range(8):
    r = [buf[w * 8 + k] for k in range(8)]
    n = max(r[7], 1)
    print(json.dumps({"wave": w, "role": "dK" if w < 4 else "dV", "steps": n, "first+soft": round(r[0] / n), "second": round(r[1] / n),
                      "dma_wait": round(r[2] / n), "barrier": round(r[3] / n), "total": round(sum(r[:4]) / n)}))
