"""Generate the BACKWARD golden vectors under tests/golden/ (run ONCE, in the authoring container).

    python tests/golden/gen_golden_bwd.py

Same arrangement as gen_golden.py: the reference's own kernels -- bwd_D_kernel
(src/flash_attention_kernels.py:115-166), bwd_kernel (:174-334) and (attempted, see case()) bwd_deterministic_kernel (:343-496) --
are executed on CPU tensors under TRITON_INTERPRET=1, called directly (`.fn[grid]`) with an explicit
(B_r, B_c) because the autotuner needs a GPU; the launch mirrors src/flash_attention_torch.py:104-155.
Under the interpreter the programs of a grid run one after the other, so the dQ lock of bwd_kernel is never
contended.  The second source of
truth is torch autograd through fp64 SDPA(scale=1) on the same (already rounded) inputs.  Only data is written.
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
TL_DTYPE = {torch.float32: tl.float32, torch.float16: tl.float16}


def run_ref_fwd(Q, K, V, B_r, B_c):
    B, H, N, d = Q.shape
    O = torch.empty(B, H, N, d, dtype=Q.dtype)
    L = torch.empty(B, H, N, 1, dtype=Q.dtype)
    fk.fwd_kernel.fn[(N // B_r, B, H)](
        Q, K, V, O, L, *Q.stride(), *K.stride(), *V.stride(), *O.stride(), L.stride(0), L.stride(1),
        B, H, N, d, TL_DTYPE[Q.dtype], B_c=B_c, B_r=B_r)
    return O, L


def run_ref_bwd(Q, K, V, O, dO, L, B_r, B_c, deterministic):
    """Mirror of flash_attention_torch.py:101-155 (and :241-292 for the deterministic class)."""
    B, H, N, d = Q.shape
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    D = torch.empty_like(L)
    n = N // 8
    lock = torch.zeros(B, H, n, dtype=torch.int32)
    written = torch.zeros(B, H, n, dtype=torch.int32)
    dt = TL_DTYPE[Q.dtype]
    fk.bwd_D_kernel.fn[(N // B_r, B, H)](O, dO, D, *O.stride(), *dO.stride(), D.stride(0), D.stride(1),
                                         B, H, N, d, dt, B_r=B_r, B_c=B_c)
    strides = (*Q.stride(), *K.stride(), *V.stride(), *dQ.stride(), *dK.stride(), *dV.stride(), *dO.stride(),
               L.stride(0), L.stride(1), D.stride(0), D.stride(1))
    if deterministic:
        fk.bwd_deterministic_kernel.fn[(N // B_c, B, H)](
            Q, K, V, dQ, dK, dV, dO, L, D, written, *strides, written.stride(0), written.stride(1),
            B, H, N, d, dt, B_c=B_c, B_r=B_r)
    else:
        fk.bwd_kernel.fn[(N // B_c, B, H)](
            Q, K, V, dQ, dK, dV, dO, L, D, lock, written, *strides, lock.stride(0), lock.stride(1),
            written.stride(0), written.stride(1), B, H, N, d, dt, B_c=B_c, B_r=B_r)
    return dQ, dK, dV, D


def autograd64(Q, K, V, dO, causal=False):
    q, k, v = (t.double().requires_grad_(True) for t in (Q, K, V))
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1, is_causal=causal)
    return torch.autograd.grad(o, (q, k, v), dO.double())


def bits(t):
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)
    return t.contiguous().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def case(name, shape, seed, dtype, tile, spread=1.0):
    torch.manual_seed(seed)
    Q, K, V, dO = ((torch.randn(*shape) * spread).to(dtype) for _ in range(4))
    O, L = run_ref_fwd(Q, K, V, *tile)
    out = dict(Q=bits(Q), K=bits(K), V=bits(V), dO=bits(dO), O_ref=bits(O), L_ref=bits(L))
    # Only bwd_kernel is run.  bwd_deterministic_kernel (same arithmetic, ordered dQ hand-off) does not terminate
    # under the interpreter even with a single key block per (b, h) -- its `while tl.load(written) != j` spin
    # (kernels.py:468) never exits here, and with more than one key block it cannot exit anywhere: program 0
    # leaves the word at 2 (xchg to 1, then +1: kernels.py:474-482) while program 1 waits for 1.
    dQ, dK, dV, D = run_ref_bwd(Q, K, V, O, dO, L, *tile, False)
    out.update(dQ_ref=bits(dQ), dK_ref=bits(dK), dV_ref=bits(dV), D_ref=bits(D))
    g = autograd64(Q, K, V, dO)
    out.update(dQ_sdpa=g[0].float().numpy(), dK_sdpa=g[1].float().numpy(), dV_sdpa=g[2].float().numpy())
    save(name, **out)
    for k in ("dQ", "dK", "dV"):
        e = np.abs(out[f"{k}_ref"].astype(np.float64) - out[f"{k}_sdpa"]).max()
        print(f"   {k}: |reference kernel - autograd64| = {e:.3e}")


def main():
    # the reference's own gradcheck shape (src/test_torch.py:4-7): B2 H2 N32 d128 fp32, seed 5
    case("bwd_test_torch_f32_seed5", (2, 2, 32, 128), 5, torch.float32, (16, 16))
    # BASELINE.json configs[0] shape, two tile shapes' worth of accumulation order
    case("bwd_c1_f32_seed11", (1, 2, 128, 64), 11, torch.float32, (32, 64))
    # fp16: P, dS are rounded to fp16 and the dots accumulate in fp16 (out_dtype=..., kernels.py:287-293)
    case("bwd_c1_f16_seed12", (1, 2, 128, 64), 12, torch.float16, (32, 32), spread=0.5)
    # [ext] causal / bf16: autograd through fp64 SDPA only
    for name, dtype, seed in (("bwd_c1_f32_causal_seed13", torch.float32, 13), ("bwd_c1_bf16_seed14", torch.bfloat16, 14)):
        torch.manual_seed(seed)
        Q, K, V, dO = (torch.randn(1, 2, 128, 64).to(dtype) for _ in range(4))
        out = dict(Q=bits(Q), K=bits(K), V=bits(V), dO=bits(dO))
        for causal in (False, True):
            g = autograd64(Q, K, V, dO, causal)
            sfx = "_causal" if causal else ""
            out.update({f"dQ_sdpa{sfx}": g[0].float().numpy(), f"dK_sdpa{sfx}": g[1].float().numpy(),
                        f"dV_sdpa{sfx}": g[2].float().numpy()})
        save(name, **out)


if __name__ == "__main__":
    main()
