#!/usr/bin/env python3
"""Quick device check of the generated assembly kernel (variant a64) against fp32 SDPA and the default kernel."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import flash_attention_dlrs_amd as fa

dev = torch.device("cuda:0")
ok = True
cases = [(1, 1, 256, False), (1, 2, 512, False), (2, 8, 1024, False), (3, 5, 768, False), (1, 16, 4096, False)]
if len(sys.argv) > 1 and sys.argv[1] == "causal":
    cases += [(1, 1, 256, True), (1, 2, 512, True), (2, 8, 1024, True), (3, 5, 768, True), (1, 16, 4096, True)]
for dtype in (torch.bfloat16, torch.float16):
    for (B, H, N, causal) in cases:
        torch.manual_seed(B * 1000 + N)
        Q, K, V = (torch.randn(B, H, N, 128, device=dev).to(dtype) for _ in range(3))
        O, L = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="a64")
        torch.cuda.synchronize()
        O2, L2 = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="mfma16h")
        ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=causal)
        e1 = (O.float() - ref).abs().max().item()
        e2 = (O2.float() - ref).abs().max().item()
        eL = (L.float() - L2.float()).abs().max().item()
        nan = torch.isnan(O.float()).sum().item()
        good = nan == 0 and e1 <= (5e-2 if dtype == torch.bfloat16 else 6e-3) and eL <= 0.51
        ok &= good
        print(json.dumps({"dtype": str(dtype), "B": B, "H": H, "N": N, "causal": causal, "err_a64": e1, "err_16h": e2, "dL": eL, "nan": nan, "ok": good}), flush=True)
        # determinism
        O3, _ = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="a64")
        torch.cuda.synchronize()
        if not torch.equal(O3, O):
            print("NOT DETERMINISTIC", (O3.float() - O.float()).abs().max().item(), flush=True)
            ok = False
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
