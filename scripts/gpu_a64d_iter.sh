#!/bin/bash
# one development iteration of the generated kernel at head size 64 (a64d): parity against fp32 SDPA and fa2_mfma16h, then A/B
set -u
cd "$(dirname "$0")/.."
timeout -k 10 300 python - <<'P' || exit 2
import torch, flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
for dtype, tol in ((torch.bfloat16, 5e-2), (torch.float16, 6e-3)):
    for causal in (False, True):
        for shape in ((1, 2, 256, 64), (2, 3, 512, 64), (1, 5, 1024, 64), (3, 40, 768, 64), (8, 16, 4096, 64)):
            g = torch.Generator().manual_seed(shape[2])
            Q, K, V = (torch.randn(*shape, generator=g).to(dtype).to(dev) for _ in range(3))
            O, L = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="a64d")
            O2, L2 = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="mfma16h")
            ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=causal)
            err, err2 = (O.float() - ref).abs().max().item(), (O2.float() - ref).abs().max().item()
            print(f"{dtype} causal={causal} {shape}: a64d max|O-sdpa| {err:.3e} (mfma16h {err2:.3e})  L diff {(L.float()-L2.float()).abs().max().item():.3e}")
            assert err <= tol and torch.isfinite(O.float()).all()
print("A64D_PARITY_OK")
P
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 20 --pairs ${PAIRS:-d64_long:auto,d64_long:a64d,d64_long_causal:auto,d64_long_causal:a64d,d64_8k:auto,d64_8k:a64d,d64_8k_causal:auto,d64_8k_causal:a64d,d64_2k:auto,d64_2k:a64d,d64_2k_causal:auto,d64_2k_causal:a64d} 2>&1 | grep pair
