#!/bin/bash
# one development iteration of the 16x16x32 form of the generated kernel: quick parity against fp32 SDPA on the device, then A/B
# against a64 in one process
set -u
cd "$(dirname "$0")/.."
timeout -k 10 300 python - <<'P' || exit 2
import torch, flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
import os
causal_list = [False] + ([True] if os.environ.get("A16_CAUSAL") else [])
for dtype, tol in ((torch.bfloat16, 5e-2), (torch.float16, 6e-3)):
    for causal in causal_list:
        for shape in ((1, 2, 256, 128), (2, 3, 512, 128), (1, 5, 1024, 128), (4, 32, 4096, 128)):
            g = torch.Generator().manual_seed(shape[2])
            Q, K, V = (torch.randn(*shape, generator=g).to(dtype).to(dev) for _ in range(3))
            O, L = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="a16")
            O2, L2 = fa.flash_attention_forward(Q, K, V, dev, causal=causal, variant="a64")
            ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=causal)
            err, err2 = (O.float() - ref).abs().max().item(), (O2.float() - ref).abs().max().item()
            same = (O == O2).float().mean().item()
            print(f"{dtype} causal={causal} {shape}: a16 max|O-sdpa| {err:.3e} (a64 {err2:.3e}) bit-equal to a64 {same:.4f}  L diff {(L.float()-L2.float()).abs().max().item():.3e}")
            assert err <= tol and torch.isfinite(O.float()).all()
print("A16_PARITY_OK")
P
timeout -k 10 300 python benchmarks/variants.py --rounds 7 --iters 20 --pairs ${PAIRS:-c3_noncausal:a64,c3_noncausal:a16,c4_per_gpu:a64,c4_per_gpu:a16,ref_bench:a64,ref_bench:a16,n2048:a64,n2048:a16} 2>&1 | grep pair
