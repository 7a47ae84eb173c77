"""ctypes front-end of oracle/fa2_oracle.c plus an independent fp64 numpy restatement.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from flash_attention_dlrs_amd/.

References (all /root/reference/src):
  flash_attention_kernels.py:17-109   the algorithm restated by fa2_oracle.c
  test_correctness.py:33              the second oracle: SDPA(Q, K, V, scale=1)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfa2_oracle.so")

DT_F32, DT_F16, DT_BF16, DT_F8E5M2, DT_F8E4M3, DT_F64 = range(6)
DTYPE_NAMES = {"float32": DT_F32, "float16": DT_F16, "bfloat16": DT_BF16,
               "float8_e5m2": DT_F8E5M2, "float8_e4m3fn": DT_F8E4M3, "float64": DT_F64}

_lib = None


def build(force=False):
    """Compile fa2_oracle.c with gcc (seconds).  FA2_ORACLE_LIB points the tests at another build of the same sources (the
    sanitizer build of `make -C oracle asan`)."""
    global _SO
    if os.environ.get("FA2_ORACLE_LIB"):
        _SO = os.environ["FA2_ORACLE_LIB"]
        return _SO
    srcs = [os.path.join(_HERE, f) for f in ("fa2_oracle.c", "fa2_oracle_bwd.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libfa2_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        i64p = ctypes.POINTER(ctypes.c_int64)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.fa2_oracle_fwd.restype = ctypes.c_int
        _lib.fa2_oracle_fwd.argtypes = [fp, fp, fp, fp, fp, i64p, i64p, i64p, i64p, i64p] + \
            [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.fa2_oracle_fwd_deferred.restype = ctypes.c_int
        _lib.fa2_oracle_fwd_deferred.argtypes = [fp, fp, fp, fp, fp, i64p, i64p, i64p, i64p, i64p] + \
            [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.fa2_oracle_fwd_f64.restype = ctypes.c_int
        _lib.fa2_oracle_fwd_f64.argtypes = [dp, dp, dp, dp, dp, i64p, i64p, i64p, i64p, i64p] + \
            [ctypes.c_int] * 5 + [ctypes.c_double]
        _lib.fa2_oracle_bwd_D.restype = ctypes.c_int
        _lib.fa2_oracle_bwd_D.argtypes = [fp, fp, fp] + [ctypes.c_int] * 5
        _lib.fa2_oracle_bwd.restype = ctypes.c_int
        _lib.fa2_oracle_bwd.argtypes = [fp] * 9 + [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        _lib.fa2_oracle_round.restype = ctypes.c_float
        _lib.fa2_oracle_round.argtypes = [ctypes.c_float, ctypes.c_int]
    return _lib


def _strides(a):
    return (ctypes.c_int64 * a.ndim)(*[s // a.itemsize for s in a.strides])


def forward(Q, K, V, dtype="float32", causal=False, scale=1.0, B_r=16, B_c=16):
    """Run the C restatement.  Q, K, V: numpy arrays (B, H, N, d) holding values already rounded to
    `dtype` (any strides).  Returns (O, L) as float32 (float64 for dtype="float64") numpy arrays whose
    values are rounded to `dtype` exactly as the reference stores them (kernels.py:107-108)."""
    dt = DTYPE_NAMES[dtype] if isinstance(dtype, str) else int(dtype)
    B, H, N, d = Q.shape
    assert K.shape == Q.shape and V.shape == Q.shape
    l = lib()
    if dt == DT_F64:
        Q, K, V = (np.asarray(x, dtype=np.float64) for x in (Q, K, V))
        O = np.empty((B, H, N, d), np.float64)
        L = np.empty((B, H, N, 1), np.float64)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        rc = l.fa2_oracle_fwd_f64(p(Q), p(K), p(V), p(O), p(L), _strides(Q), _strides(K), _strides(V),
                                  _strides(O), (ctypes.c_int64 * 2)(H * N, N), B, H, N, d,
                                  int(bool(causal)), float(scale))
    else:
        Q, K, V = (np.asarray(x, dtype=np.float32) for x in (Q, K, V))
        O = np.empty((B, H, N, d), np.float32)
        L = np.empty((B, H, N, 1), np.float32)
        B_r = min(B_r, N)
        B_c = min(B_c, N)
        p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        rc = l.fa2_oracle_fwd(p(Q), p(K), p(V), p(O), p(L), _strides(Q), _strides(K), _strides(V),
                              _strides(O), (ctypes.c_int64 * 2)(H * N, N), B, H, N, d, dt,
                              int(bool(causal)), float(scale), B_r, B_c)
    if rc != 0:
        raise ValueError(f"fa2_oracle_fwd rc={rc} (N={N} must be a multiple of B_r={B_r}, B_c={B_c})")
    return O, L


def forward_deferred(Q, K, V, dtype, causal=False, scale=1.0, G=32, B_c=64, thr=60.0, sum_rounded=True, ceil_m=False):
    """fa2_oracle_fwd_deferred: the restatement with the MFMA kernels' deferred running maximum (per G-row group, threshold
    thr in log2 units), the single-rounding exp2(fma(S, c, -m)) and row sums of the rounded P.  Any N."""
    dt = DTYPE_NAMES[dtype] if isinstance(dtype, str) else int(dtype)
    B, H, N, d = Q.shape
    Q, K, V = (np.asarray(x, dtype=np.float32) for x in (Q, K, V))
    O = np.empty((B, H, N, d), np.float32)
    L = np.empty((B, H, N, 1), np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    rc = lib().fa2_oracle_fwd_deferred(p(Q), p(K), p(V), p(O), p(L), _strides(Q), _strides(K), _strides(V), _strides(O),
                                       (ctypes.c_int64 * 2)(H * N, N), B, H, N, d, dt, int(bool(causal)), float(scale),
                                       int(G), int(B_c), float(thr), int(bool(sum_rounded)), int(bool(ceil_m)))
    if rc != 0:
        raise ValueError(f"fa2_oracle_fwd_deferred rc={rc}")
    return O, L


def backward(Q, K, V, O, dO, L, dtype="float32", causal=False, scale=1.0, B_r=16, B_c=16):
    """The C restatement of the reference's bwd_D_kernel + bwd_kernel (fa2_oracle_bwd.c; kernels.py:115-334).
    Inputs hold values already rounded to `dtype`; L is the forward's log2-domain LSE, shape (B, H, N[, 1]).
    Returns (dQ, dK, dV, D) as float32 arrays whose values are rounded to `dtype` as the reference stores them."""
    dt = DTYPE_NAMES[dtype] if isinstance(dtype, str) else int(dtype)
    B, H, N, d = Q.shape
    c = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    Q, K, V, O, dO = (c(x) for x in (Q, K, V, O, dO))
    L = c(np.asarray(L).reshape(B, H, N))
    dQ, dK, dV = (np.empty((B, H, N, d), np.float32) for _ in range(3))
    D = np.empty((B, H, N), np.float32)
    p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    l = lib()
    l.fa2_oracle_bwd_D(p(O), p(dO), p(D), B, H, N, d, dt)
    rc = l.fa2_oracle_bwd(p(Q), p(K), p(V), p(dO), p(L), p(D), p(dQ), p(dK), p(dV), B, H, N, d, dt,
                          int(bool(causal)), float(scale), min(B_r, N), min(B_c, N))
    if rc != 0:
        raise ValueError(f"fa2_oracle_bwd rc={rc} (N={N} must be a multiple of B_r={B_r}, B_c={B_c})")
    return dQ, dK, dV, D


def grads_f64(Q, K, V, dO, causal=False, scale=1.0):
    """Independent fp64 numpy restatement of the gradients of softmax(scale * Q K^T [+ mask]) V."""
    Q, K, V, dO = (np.asarray(x, dtype=np.float64) for x in (Q, K, V, dO))
    S = np.einsum("bhnd,bhmd->bhnm", Q, K) * scale
    if causal:
        N = Q.shape[2]
        S = np.where(np.tril(np.ones((N, N), bool)), S, -np.inf)
    P = np.exp(S - S.max(axis=-1, keepdims=True))
    P /= P.sum(axis=-1, keepdims=True)
    dV = np.einsum("bhnm,bhnd->bhmd", P, dO)
    dP = np.einsum("bhnd,bhmd->bhnm", dO, V)
    D = (P * dP).sum(axis=-1, keepdims=True)
    dS = P * (dP - D) * scale
    return np.einsum("bhnm,bhmd->bhnd", dS, K), np.einsum("bhnm,bhnd->bhmd", dS, Q), dV


def round_scalar(x, dtype):
    dt = DTYPE_NAMES[dtype] if isinstance(dtype, str) else int(dtype)
    return lib().fa2_oracle_round(float(x), dt)


def sdpa_f64(Q, K, V, causal=False, scale=1.0):
    """Independent restatement in numpy float64: softmax(scale * Q K^T [+ causal mask]) V and the
    log2-domain log-sum-exp L.  Mirrors test_correctness.py:33."""
    Q, K, V = (np.asarray(x, dtype=np.float64) for x in (Q, K, V))
    S = np.einsum("bhnd,bhmd->bhnm", Q, K) * scale
    if causal:
        N = Q.shape[2]
        S = np.where(np.tril(np.ones((N, N), bool)), S, -np.inf)
    mx = S.max(axis=-1, keepdims=True)
    P = np.exp(S - mx)
    l = P.sum(axis=-1, keepdims=True)
    O = np.einsum("bhnm,bhmd->bhnd", P / l, V)
    L = (mx + np.log(l)) * np.log2(np.e)
    return O, L
