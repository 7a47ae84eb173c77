"""Generate the golden vectors under tests/golden/ (run ONCE, in the authoring container).

    python tests/golden/gen_golden.py

Needs /root/reference (read-only) and Triton's CPU interpreter; neither exists on the GPU box, which
only ever sees the committed .npz files.  Two sources of truth are recorded per case:

  * O_ref / L_ref : the reference's own fwd_kernel (src/flash_attention_kernels.py:17-109) executed on
    CPU tensors with TRITON_INTERPRET=1.  The autotuner (kernels.py:11-15) needs a GPU benchmarker, so
    the jitted function is called directly (`fwd_kernel.fn[grid]`) with an explicit (B_r, B_c) taken from
    the reference's config list; `autotune_configs.is_cuda` is overridden IN MEMORY because
    autotune_configs.py:197-201 refuses to import without a CUDA driver.  Reference files are untouched.
  * O_sdpa : torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1 [, is_causal]) -- the
    oracle the reference's own test uses (src/test_correctness.py:33) -- evaluated in fp64 on the
    (already rounded) inputs so it is the exact target for every dtype.

Only data is written: inputs and expected outputs.
"""
import os
import sys

os.environ["TRITON_INTERPRET"] = "1"
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch
import triton.language as tl

import autotune_configs

autotune_configs.is_cuda = lambda: True
import flash_attention_kernels as fk  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TL_DTYPE = {torch.float32: tl.float32, torch.float16: tl.float16, torch.float8_e5m2: tl.float8e5}


def next_pow2(x):
    return 1 << (x - 1).bit_length()


def run_ref_kernel(Q, K, V, B_r, B_c):
    """Mirror of the launch in flash_attention_wrappers.py:37-61 on CPU tensors."""
    B, H, N, d = Q.shape
    O = torch.empty(B, H, N, d, dtype=Q.dtype)
    L = torch.empty(B, H, N, 1, dtype=Q.dtype)
    fk.fwd_kernel.fn[(N // B_r, B, H)](
        Q, K, V, O, L, *Q.stride(), *K.stride(), *V.stride(), *O.stride(), L.stride(0), L.stride(1),
        B, H, N, d, TL_DTYPE[Q.dtype], B_c=B_c, B_r=B_r)
    return O, L


def sdpa64(Q, K, V, causal=False):
    return torch.nn.functional.scaled_dot_product_attention(
        Q.double(), K.double(), V.double(), scale=1, is_causal=causal)


def bits(t):
    """Lossless numpy view of a torch tensor of any float dtype."""
    if t.dtype in (torch.float32, torch.float64, torch.float16):
        return t.contiguous().numpy()
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().view(torch.uint8).numpy()  # fp8


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def randn(shape, seed, dtype=torch.float32):
    torch.manual_seed(seed)
    return tuple(torch.randn(*shape).to(dtype) for _ in range(3))


def main():
    # (1) c1 = BASELINE.json configs[0]: B1 H2 N128 d64 fp32, the reference's CPU-runnable case.
    for seed, tiles in ((0, ((16, 16), (32, 64))), (1, ((64, 32),))):
        Q, K, V = randn((1, 2, 128, 64), seed)
        out = dict(Q=bits(Q), K=bits(K), V=bits(V), O_sdpa=sdpa64(Q, K, V).float().numpy())
        for (br, bc) in tiles:
            O, L = run_ref_kernel(Q, K, V, br, bc)
            out[f"O_ref_{br}x{bc}"] = bits(O)
            out[f"L_ref_{br}x{bc}"] = bits(L)
        save(f"c1_f32_seed{seed}", **out)

    # (2) fp16 at the c1 shape: reference kernel stores O and L in fp16 (kernels.py:107-108).
    Q, K, V = randn((1, 2, 128, 64), 2, torch.float16)
    O, L = run_ref_kernel(Q, K, V, 32, 32)
    save("c1_f16_seed2", Q=bits(Q), K=bits(K), V=bits(V), O_ref_32x32=bits(O), L_ref_32x32=bits(L),
         O_sdpa=sdpa64(Q, K, V).float().numpy())

    # (3) [ext] causal and bf16: the reference kernel has neither (torch.py:18 raises for bf16), so
    # SDPA(scale=1, is_causal) on the rounded inputs is the only truth here.
    Q, K, V = randn((1, 2, 128, 64), 3)
    save("c1_f32_causal_seed3", Q=bits(Q), K=bits(K), V=bits(V),
         O_sdpa=sdpa64(Q, K, V, causal=True).float().numpy())
    Q, K, V = randn((1, 2, 128, 64), 4, torch.bfloat16)
    save("c1_bf16_seed4", Q=bits(Q), K=bits(K), V=bits(V),
         O_sdpa=sdpa64(Q, K, V).float().numpy(),
         O_sdpa_causal=sdpa64(Q, K, V, causal=True).float().numpy())

    # (4) padding paths: d=40 -> 64 and d=8 -> 16 (flash_attention_torch.py:38-47).  The kernel is
    # run on the zero-padded tensors exactly as the host glue would; expected O is the [:d] slice.
    for (shape, seed, tile) in (((1, 2, 64, 40), 5, (32, 32)), ((1, 1, 32, 8), 6, (16, 16))):
        Q, K, V = randn(shape, seed)
        d = shape[-1]
        dp = max(next_pow2(d), 16)
        pad = lambda t: torch.nn.functional.pad(t, (0, dp - d))
        O, L = run_ref_kernel(pad(Q), pad(K), pad(V), *tile)
        save(f"pad_d{d}_f32_seed{seed}", Q=bits(Q), K=bits(K), V=bits(V), O_ref=bits(O[..., :d]),
             L_ref=bits(L), O_sdpa=sdpa64(Q, K, V).float().numpy())

    # (5) non-contiguous inputs: (B, N, H, d) storage viewed as (B, H, N, d) (kernels.py:45-79 honour
    # strides).  Stored contiguous in storage order; the test re-creates the transposed view.
    torch.manual_seed(7)
    Qs, Ks, Vs = (torch.randn(1, 64, 2, 32) for _ in range(3))
    Q, K, V = (t.transpose(1, 2) for t in (Qs, Ks, Vs))
    O, L = run_ref_kernel(Q, K, V, 16, 32)
    save("strided_bnhd_f32_seed7", Q_storage=bits(Qs), K_storage=bits(Ks), V_storage=bits(Vs),
         O_ref=bits(O), L_ref=bits(L), O_sdpa=sdpa64(Q, K, V).float().numpy())

    # (6) smallest supported N (16) and a non-power-of-two multiple of 16 (48) (autotune_configs.py:176-187).
    for (shape, seed) in (((1, 1, 16, 16), 8), ((1, 2, 48, 32), 9)):
        Q, K, V = randn(shape, seed)
        O, L = run_ref_kernel(Q, K, V, 16, 16)
        save(f"n{shape[2]}_f32_seed{seed}", Q=bits(Q), K=bits(K), V=bits(V), O_ref=bits(O), L_ref=bits(L),
             O_sdpa=sdpa64(Q, K, V).float().numpy())

    # (7) fp8 e5m2, the reference's only fp8 (flash_attention_torch.py:14-15).  Raw bytes are stored.
    Q, K, V = randn((1, 2, 64, 32), 10, torch.float8_e5m2)
    O, L = run_ref_kernel(Q, K, V, 16, 32)
    save("f8e5m2_seed10", Q=bits(Q), K=bits(K), V=bits(V), O_ref=bits(O), L_ref=bits(L),
         O_sdpa=sdpa64(Q.float(), K.float(), V.float()).float().numpy())


def main_d128():
    """(8) round 3: the head size of the north-star kernel.  The default f16/bf16 kernel `a64` needs d = 128 and N >= 256, and
    fp32 at d = 128 is the reference test's own head size (src/test_correctness.py:9-14) -- no vector above covers either.
    Run separately (`python tests/golden/gen_golden.py d128`) so the fixtures above are not rewritten."""
    Q, K, V = randn((1, 1, 512, 128), 15, torch.float16)     # two 256-row jobs of the a64 kernel, eight 64-key tiles
    O, L = run_ref_kernel(Q, K, V, 64, 64)
    save("d128_f16_n512_seed15", Q=bits(Q), K=bits(K), V=bits(V), O_ref_64x64=bits(O), L_ref_64x64=bits(L),
         O_sdpa=sdpa64(Q, K, V).float().numpy())
    Q, K, V = randn((1, 1, 256, 128), 16)
    O, L = run_ref_kernel(Q, K, V, 32, 32)
    save("d128_f32_n256_seed16", Q=bits(Q), K=bits(K), V=bits(V), O_ref_32x32=bits(O), L_ref_32x32=bits(L),
         O_sdpa=sdpa64(Q, K, V).float().numpy())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "d128":
        main_d128()
    else:
        main()
