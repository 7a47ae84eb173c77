"""Runs a generated kernel on the emulator for one problem (test infrastructure, see emu.py)."""
from __future__ import annotations

import struct

import numpy as np

from .emu import Memory, Workgroup, bf16_to_f32, f32_to_bf16_rne
from .fa2_a64_gen import KARG_SIZE, Gen

LOG2E = 1.4426950408889634
# deferral threshold of the running maximum (log2 units) the launcher passes (fa2_a64.hip): P <= 2^thr; f16 P must stay below 65 504
A64_THR = {"bf16": 60.0, "f16": 15.875, "e4m3": 8.5, "e5m2": 15.0}     # (fp8: P <= 2^8.5 inside e4m3's 448, 2^15 inside e5m2's 57 344)
ESIZE = {"bf16": 2, "f16": 2, "e4m3": 1, "e5m2": 1}


def _f8(dtype):
    import torch
    return torch.float8_e4m3fn if dtype == "e4m3" else torch.float8_e5m2


def to_dt(x, dtype):
    if dtype == "bf16":
        return f32_to_bf16_rne(np.asarray(x, np.float32))
    if dtype in ("e4m3", "e5m2"):
        import torch
        return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(_f8(dtype)).view(torch.uint8).numpy()
    return np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def from_dt(u16, dtype):
    if dtype == "bf16":
        return bf16_to_f32(u16)
    if dtype in ("e4m3", "e5m2"):
        import torch
        return torch.from_numpy(np.ascontiguousarray(u16, np.uint8)).view(_f8(dtype)).float().numpy()
    return u16.view(np.float16).astype(np.float32)


def pack_kargs(ptrs, strides_bh, strides_n, N, H, nq, total, c, thr, nunit, G, nbh, nwg, dbg=0, pow2=True, pairs=False):
    """ptrs: Q,K,V,O,L addresses; strides_bh: (qs_b, qs_h, ks_b, ks_h, vs_b, vs_h, os_b, os_h, ls_b, ls_h) in bytes;
    strides_n: (qs_n, ks_n, vs_n, os_n) in bytes -- the layout of fa2_a64_gen.k_setup"""
    b = struct.pack("<5Q", *ptrs) + struct.pack("<10q", *strides_bh) + struct.pack("<4i", *strides_n)
    b += struct.pack("<4i", N, H, nq, total) + struct.pack("<2f2i", c, thr, nunit, G) + struct.pack("<2i", nbh, nwg)
    b += struct.pack("<Q", dbg)
    lg = 0
    ispow2 = lambda x: x > 0 and (x & (x - 1)) == 0
    if pow2 and nbh % 8 == 0 and ispow2(H) and ispow2(G) and ispow2(G * nunit):
        lg = (H.bit_length() - 1) | ((G.bit_length() - 1) << 8) | (((G * nunit).bit_length() - 1) << 16) | (1 << 24)
    if pairs:      # causal: the light job of a unit walks its non-diagonal key tiles downwards (fa2_a64.hip)
        lg |= 1 << 25
    b += struct.pack("<II", lg, 0)
    assert len(b) == KARG_SIZE, len(b)
    return b


def run(prog, Q, K, V, dtype="bf16", causal=False, scale=1.0, nwg=None, order=None, G=1, pow2=True, thr_override=None, pairs=False):
    """Q, K, V: float32 arrays (B, H, N, 128), rounded to dtype here.  Returns O (B,H,N,128) f32, L (B,H,N) f32."""
    B, H, N, D = Q.shape
    assert D in (64, 128)      # (128: a64, a16, a8;  64: a64d)
    mem = Memory()
    bufs = {}
    for nm, x in (("Q", Q), ("K", K), ("V", V)):
        arr = to_dt(x, dtype).view(np.uint8).reshape(-1).copy()
        bufs[nm] = (mem.alloc(arr), arr)
    es = ESIZE[dtype]
    o_arr = np.full(B * H * N * D * es, 0xAB, np.uint8)
    l_arr = np.full(B * H * N * es, 0xAB, np.uint8)
    bufs["O"] = (mem.alloc(o_arr), o_arr)
    bufs["L"] = (mem.alloc(l_arr), l_arr)
    nq = (N + 255) // 256      # (N not a multiple of 256: the ragged kernels; the buffers hold exactly N rows)
    nunit = (nq + 1) // 2 if causal else nq
    nbh = B * H
    total = nunit * nbh
    assert causal or not pairs
    nwg = nwg or min(total, 256)
    sb, sh, sn = H * N * D * es, N * D * es, D * es
    thr = A64_THR[dtype]
    if thr_override is not None:
        thr = thr_override
    ka = pack_kargs([bufs[k][0] for k in "QKVOL"], (sb, sh) * 4 + (H * N * es, N * es), (sn,) * 4, N, H, nq, total,
                    float(scale * LOG2E), thr, nunit, G, nbh, nwg, pow2=pow2, pairs=pairs)
    ka_arr = np.frombuffer(ka, np.uint8).copy()
    ka_addr = mem.alloc(ka_arr)
    steps = 0
    for wg in range(nwg):
        g = Workgroup(prog, mem, 4, wg, ka_addr)
        steps += g.run(order=order)
    view = np.uint16 if es == 2 else np.uint8
    O = from_dt(o_arr.view(view), dtype).reshape(B, H, N, D)
    L = from_dt(l_arr.view(view), dtype).reshape(B, H, N)
    return O, L, steps


def reference(Q, K, V, dtype="bf16", causal=False, scale=1.0):
    """fp64 attention on the dtype-rounded inputs (the checker for the emulated kernel)"""
    q = from_dt(to_dt(Q, dtype), dtype).astype(np.float64)
    k = from_dt(to_dt(K, dtype), dtype).astype(np.float64)
    v = from_dt(to_dt(V, dtype), dtype).astype(np.float64)
    s = np.einsum("bhnd,bhmd->bhnm", q, k) * scale
    if causal:
        N = q.shape[2]
        s = np.where(np.tril(np.ones((N, N), bool)), s, -np.inf)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    O = np.einsum("bhnm,bhmd->bhnd", p / l, v)
    L = (m[..., 0] + np.log(l[..., 0])) * LOG2E
    return O, L
