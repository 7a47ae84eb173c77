#!/bin/bash
# causal head pairs (lockstep K / V streams): parity on pair-mode shapes, then A/B against FA2_A64_PAIRS=0 in one process
set -u
cd "$(dirname "$0")/.."
timeout -k 10 300 python - <<'P' || exit 2
import torch, flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
for dtype, tol in ((torch.bfloat16, 5e-2), (torch.float16, 6e-3)):
    for shape in ((1, 16, 512, 128), (2, 8, 768, 128), (1, 16, 1280, 128), (3, 16, 2048, 128), (4, 32, 4096, 128), (1, 48, 4096, 128), (2, 40, 1024, 128)):
        g = torch.Generator().manual_seed(shape[2] + shape[1])
        Q, K, V = (torch.randn(*shape, generator=g).to(dtype).to(dev) for _ in range(3))
        O, L = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="a64")
        O2, L2 = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="mfma16h")
        ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0, is_causal=True)
        err, err2 = (O.float() - ref).abs().max().item(), (O2.float() - ref).abs().max().item()
        Lr = torch.logsumexp((Q.float() @ K.float().transpose(-1, -2)).masked_fill(~torch.ones(shape[2], shape[2], dtype=torch.bool, device=dev).tril(), float("-inf")), -1) * 1.4426950408889634
        lerr = (L.float().squeeze(-1) - Lr).abs().max().item()
        print(f"{dtype} {shape}: a64 max|O-sdpa| {err:.3e} (mfma16h {err2:.3e}) max|L-ref| {lerr:.3e}")
        assert err <= tol and torch.isfinite(O.float()).all()
print("PAIRS_PARITY_OK")
P
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_exp.so
timeout -k 10 300 python benchmarks/variants.py --rounds 9 --iters 20 --pairs c3:a64,c3:a64:FA2_A64_PAIRS=0,causal_8k:a64,causal_8k:a64:FA2_A64_PAIRS=0,causal_2k:a64,causal_2k:a64:FA2_A64_PAIRS=0,causal_16k:a64,causal_16k:a64:FA2_A64_PAIRS=0 2>&1 | grep pair
