#!/bin/bash
# one development iteration of the generated fp8 kernel (a8): parity against fa2_mfma8x and fp32 SDPA of the fp8 inputs, then A/B
set -u
cd "$(dirname "$0")/.."
timeout -k 10 300 python - <<'P' || exit 2
import torch, flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
for dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
    for shape in ((1, 2, 256, 128), (2, 3, 512, 128), (1, 5, 1024, 128), (3, 40, 768, 128), (4, 32, 4096, 128)):
        g = torch.Generator().manual_seed(shape[2])
        Q, K, V = ((torch.randn(*shape, generator=g) * 0.5).to(dtype).to(dev) for _ in range(3))
        O, L = fa.flash_attention_forward(Q, K, V, dev, variant="a8")
        O2, L2 = fa.flash_attention_forward(Q, K, V, dev, variant="mfma8x")
        ref = torch.nn.functional.scaled_dot_product_attention(Q.float(), K.float(), V.float(), scale=1.0)
        err, err2 = (O.float() - ref).abs().max().item(), (O2.float() - ref).abs().max().item()
        same = (O.view(torch.uint8) == O2.view(torch.uint8)).float().mean().item()
        samel = (L.view(torch.uint8) == L2.view(torch.uint8)).float().mean().item()
        print(f"{dtype} {shape}: a8 max|O-sdpa| {err:.3e} (mfma8x {err2:.3e}) bit-equal to mfma8x O {same:.4f} L {samel:.4f}")
        assert torch.isfinite(O.float()).all() and err <= 2 * err2 + 0.05
print("A8_PARITY_OK")
P
timeout -k 10 300 python benchmarks/variants.py --rounds 7 --iters 10 --pairs ${PAIRS:-c5_per_gpu:mfma8x,c5_per_gpu:a8,fp8_4k:mfma8x,fp8_4k:a8,fp8_1k:mfma8x,fp8_1k:a8} 2>&1 | grep pair
