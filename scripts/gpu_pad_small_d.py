"""head sizes below 64 that are multiples of 8 (the d-predicated first MFMA kernel takes them as they are) against a host pad to 64
(the generated d = 64 kernel), pad copies and the slice of O included"""
import json, sys, torch
sys.path.insert(0, ".")
from flash_attention_dlrs_amd import flash_attention_forward
from flash_attention_dlrs_amd.flash_attention_torch import pad_last_dim
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


for dtype in (torch.bfloat16, torch.float16):
    for (B, H, N) in ((4, 32, 4096), (8, 16, 1024), (2, 8, 1024)):
        for d in (16, 32, 40, 48):
            for causal in (False, True):
                torch.manual_seed(1)
                Q, K, V = (torch.randn(B, H, N, d, device=dev).to(dtype) for _ in range(3))
                row = {"dtype": str(dtype).replace("torch.", ""), "B": B, "H": H, "N": N, "d": d, "causal": causal}
                row["as_is_ms"] = round(timed(lambda: flash_attention_forward(Q, K, V, dev, causal=causal, variant="mfma16_w8" if B * H * N >= 131072 else "mfma16")), 4)

                def run():
                    O, L = flash_attention_forward(pad_last_dim(Q, 64), pad_last_dim(K, 64), pad_last_dim(V, 64), dev, causal=causal)
                    return O[..., :d]
                row["pad64_ms"] = round(timed(run), 4)
                print(json.dumps(row), flush=True)
