"""torch surface of the FA-2 forward -- counterpart of the reference's src/flash_attention_torch.py.

Same names, arity, checks, exceptions, padding and return values as the reference:
  MIN_TENSOR_SIZE, convert_triton_dtype           (reference torch.py:5-18)
  FlashAttention, FlashAttentionDeterministic      (reference torch.py:21-158, :161-294)
The forward launches the hand-written gfx950 kernel through the C ABI (include/fa2_fwd.h) instead
of Triton.  `causal` and `scale` are optional extra positional arguments of `.apply` with
reference-preserving defaults (no mask, scale 1).

The backward launches the hand-written gfx950 backward kernels through
include/fa2_bwd.h: D = rowsum(dO * O), then a key-block-owner kernel for dK / dV and a query-block-owner kernel
for dQ -- no cross-workgroup sum, hence deterministic, so FlashAttention and FlashAttentionDeterministic share
it (the reference's two classes differ only in how they serialise the dQ sum, torch.py:86-158 vs :226-294).
"""
import math

import torch

from . import _lib, autotune

MIN_TENSOR_SIZE = 16

_DTYPE_MAP = {
    torch.float64: _lib.FA2_DTYPE_F64,
    torch.float32: _lib.FA2_DTYPE_F32,
    torch.float16: _lib.FA2_DTYPE_F16,
    torch.float8_e5m2: _lib.FA2_DTYPE_F8E5M2,
    # extensions (BASELINE.json configs c3-c5); the reference raises TypeError for these
    torch.bfloat16: _lib.FA2_DTYPE_BF16,
    torch.float8_e4m3fn: _lib.FA2_DTYPE_F8E4M3,
}


def convert_triton_dtype(torch_dtype):
    """torch dtype -> kernel dtype enum (include/fa2_fwd.h FA2_DTYPE_*).  Name kept from the
    reference (torch.py:7-18), where it returns a triton dtype; TypeError for anything unsupported."""
    try:
        return _DTYPE_MAP[torch_dtype]
    except KeyError:
        raise TypeError(f"dtype {torch_dtype} not supported.") from None


def next_power_of_2(n):
    return 1 << (int(n) - 1).bit_length() if n > 1 else 1


def pad_last_dim(t, d_proper):
    """Zero-pad the last dimension to d_proper (reference torch.py:40-45).  fp8 tensors are padded
    through their byte view (0x00 is +0.0 in both fp8 formats)."""
    d = t.shape[-1]
    if d == d_proper:
        return t
    if t.dtype in (torch.float8_e5m2, torch.float8_e4m3fn):
        out = torch.zeros(*t.shape[:-1], d_proper, dtype=torch.uint8, device=t.device)
        out[..., :d] = t.view(torch.uint8)
        return out.view(t.dtype)
    return torch.nn.functional.pad(t, (0, d_proper - d), mode="constant", value=0.0)


def forward_head_size(dtype, B, H, N, d, causal=False):
    """Head size the forward kernels are RUN at.  The kernels take any d (the reference pads every d that is not a power of two,
    torch.py:38-47), but only some on the matrix cores: f16 / bf16 multiples of 8 and fp32 multiples of 4 up to 128.  Everything
    else would run on the VALU kernel, 60-90 times slower than a zero-padded launch (profiles/r03/pad_vs_predicated.jsonl: bf16
    B4 H32 N4096 d = 100: 87.5 ms as it is, 0.98 ms padded to 128, the three pad copies and the slice of O included).  So:
      * f16 / bf16, d not a multiple of 8: pad to 64 (d < 64) or 128 -- the pipelined kernels; as fast as or faster than the next
        multiple of 8 on every shape measured;
      * f16 / bf16, 64 < d < 128 a multiple of 8: the d-predicated kernel as it is on small grids (29 vs 45 us at B2 H8 N1024), padded
        to 128 from 128 Ki rows on (B4 H32 N4096 d = 96: 1.08 -> 0.90 ms); d < 64 a multiple of 8: as it is, except causal problems of
        256 Ki rows and more, which go to the generated d = 64 kernel (+13 .. 26 %, profiles/r03/pad_small_d.jsonl);
      * fp32, d not a multiple of 4: the next multiple of 4 (the predicated fp32 MFMA kernel);
      * fp8, d < 128: 128 (fp8 runs on the matrix cores at that head size only).
    Zero-padding is exact: the extra products are zeros, the extra columns of O are sliced away (torch.py:81-82)."""
    if dtype in (torch.float16, torch.bfloat16) and d <= 128:
        if d % 8:
            return 64 if d < 64 else 128
        if 64 < d < 128 and B * H * N >= 131072:
            return 128
        if d < 64 and causal and B * H * N >= 262144:
            return 64      # (multiples of 8 below 64, causal, large: the generated d = 64 kernel, 0.34 vs 0.43 ms at B4 H32 N4096 d = 16 .. 48;
            #                non-causal the two are level, and below that size the pad copies cost more than they buy)
    if dtype == torch.float32 and d <= 128 and d % 4:
        return (d + 3) // 4 * 4
    if dtype in (torch.float8_e4m3fn, torch.float8_e5m2) and d < 128:
        return 128       # (the fp8 matrix kernels exist at d = 128 only; any other head size would run on the VALU kernel)
    return d


def _check_inputs(Q, K, V):
    dev = Q.device
    if dev.type != "cuda" or dev != K.device or dev != V.device:
        raise NotImplementedError("Q, K, V must be on the same CUDA device")
    if Q.dim() != 4 or Q.shape != K.shape or Q.shape != V.shape:
        raise ValueError("Q, K, V must all be of shape (B, H, N, d)")
    if Q.dtype != K.dtype or K.dtype != V.dtype:
        raise ValueError("Q, K, V must have same dtype")


def _forward_impl(ctx, Q, K, V, causal, scale):
    _check_inputs(Q, K, V)
    B, H, N, d = Q.shape
    dtype = convert_triton_dtype(Q.dtype)

    # Non-power-of-2 d or d < 16: the reference pads Q, K, V on the host (torch.py:38-47) and returns the O[..., :d] view
    # of a padded O.  The forward kernels take any d (the MFMA kernels zero-fill the missing
    # columns on load, the generic kernel loops to d): head sizes the matrix cores take run as they are, O comes back with
    # exactly d columns; the others are padded as the reference pads them (forward_head_size).  The backward still wants a
    # power of two and pads what it was handed (_backward_impl).
    d_proper = max(next_power_of_2(d), MIN_TENSOR_SIZE)
    padded = d_proper != d
    d_run = forward_head_size(Q.dtype, B, H, N, d, causal)

    # O inherits Q's strides, L is (B, H, N, 1) in the input dtype (reference torch.py:50-51)
    L = torch.empty(B, H, N, 1, dtype=Q.dtype, device=Q.device)
    if d_run != d:
        Qr, Kr, Vr = (pad_last_dim(t, d_run) for t in (Q, K, V))
        Or = torch.empty_like(Qr)
        _lib.fa2_fwd(Qr, Kr, Vr, Or, L, dtype, causal=causal, scale=scale,
                     variant=autotune.pick(Qr, Kr, Vr, Or, L, dtype, causal, scale))
        O = Or[..., :d]        # (a view of the padded O, as the reference returns it: torch.py:81-82)
    else:
        O = torch.empty_like(Q)
        # static gfx950 tile table, or the on-box tuner's choice when FA2_AUTOTUNE=1 (autotune.py; reference:
        # the Triton autotuner keyed on (B, H, N, d), kernels.py:11-15)
        _lib.fa2_fwd(Q, K, V, O, L, dtype, causal=causal, scale=scale,
                     variant=autotune.pick(Q, K, V, O, L, dtype, causal, scale))

    ctx.save_for_backward(Q, K, V, O, L)
    ctx.padded = padded
    ctx.d_used = d_proper
    ctx.d_orig = d
    ctx.causal = bool(causal)
    ctx.scale = float(scale)
    return O


def attention_backward_recompute(Q, K, V, O, dO, L, causal=False, scale=1.0):
    """dQ, dK, dV from the saved statistics in plain torch ops: a readable restatement used by the tests only
    (the product path is _backward_native below).
    P = exp2(scale * S * log2e - L) (reference kernels.py:283-285), D = rowsum(dO * O) (kernels.py:120-166)."""
    f = torch.float64 if Q.dtype == torch.float64 else torch.float32
    q, k, v, o, do, l = (t.to(f) for t in (Q, K, V, O, dO, L))
    S = torch.matmul(q, k.transpose(-1, -2)) * (scale * math.log2(math.e))
    if causal:
        N = Q.shape[2]
        mask = torch.ones(N, N, dtype=torch.bool, device=Q.device).tril()
        S = S.masked_fill(~mask, float("-inf"))
    P = torch.exp2(S - l)
    dV = torch.matmul(P.transpose(-1, -2), do)
    dP = torch.matmul(do, v.transpose(-1, -2))
    D = (do * o).sum(dim=-1, keepdim=True)
    dS = P * (dP - D) * scale
    dQ = torch.matmul(dS, k)
    dK = torch.matmul(dS.transpose(-1, -2), q)
    return dQ.to(Q.dtype), dK.to(K.dtype), dV.to(V.dtype)


def backward_native(Q, K, V, O, dO, L, causal=False, scale=1.0, variant="auto"):
    """Host glue of the backward launch (reference torch.py:101-155): allocate dQ, dK, dV (strides of Q, K, V) and
    the scratch D, launch, return the gradients.  Inputs are already padded to a supported d."""
    dtype = convert_triton_dtype(Q.dtype)
    if Q.dtype in (torch.float8_e5m2, torch.float8_e4m3fn):
        raise TypeError(f"dtype {Q.dtype} not supported by the backward.")
    B, H, N, d = Q.shape
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    # scratch: rowsum(dO * O) and the fp32 row statistic handed from the dQ launch to the dK/dV launch (fa2_bwd.h)
    D = torch.empty(2, B, H, N, 1, dtype=torch.float64 if Q.dtype == torch.float64 else torch.float32, device=Q.device)
    if dO.stride(-1) != 1:
        dO = dO.contiguous()
    _lib.fa2_bwd(Q, K, V, O, dO, L, dQ, dK, dV, D, dtype, causal=causal, scale=scale,
                 variant=_lib.BWD_VARIANTS[variant])
    return dQ, dK, dV


def _backward_impl(ctx, dO):
    Q, K, V, O, L = ctx.saved_tensors
    if Q.dtype != dO.dtype:
        raise ValueError("dO must have same dtype as inputs")
    if ctx.padded:   # (reference torch.py:91-100 pads in its backward as well)
        Q, K, V, O, dO = (pad_last_dim(t, ctx.d_used) for t in (Q, K, V, O, dO))
    dQ, dK, dV = backward_native(Q, K, V, O, dO, L, ctx.causal, ctx.scale)
    if ctx.padded:
        d = ctx.d_orig
        return dQ[..., :d], dK[..., :d], dV[..., :d], None, None
    return dQ, dK, dV, None, None


class FlashAttention(torch.autograd.Function):
    """O = softmax(Q K^T) V with scale 1 (reference torch.py:21-84).  `FlashAttention.apply(Q, K, V)`;
    optional extras `FlashAttention.apply(Q, K, V, causal, scale)`."""

    @staticmethod
    def forward(ctx, Q, K, V, causal=False, scale=1.0):
        return _forward_impl(ctx, Q, K, V, causal, scale)

    @staticmethod
    def backward(ctx, grad_outputs, *args):
        return _backward_impl(ctx, grad_outputs)


class FlashAttentionDeterministic(torch.autograd.Function):
    """Same forward as FlashAttention (the reference's two forwards are identical, torch.py:161-224);
    the recompute backward used here is deterministic by construction."""

    @staticmethod
    def forward(ctx, Q, K, V, causal=False, scale=1.0):
        return _forward_impl(ctx, Q, K, V, causal, scale)

    @staticmethod
    def backward(ctx, grad_outputs, *args):
        return _backward_impl(ctx, grad_outputs)
