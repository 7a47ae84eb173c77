"""ctypes binding of libfa2_hip.so -- the only way the Python surface reaches the GPU.

Replaces the `fwd_kernel[grid](...)` Triton launch of the reference
(src/flash_attention_torch.py:59-74, src/flash_attention_wrappers.py:46-61).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FA2_HIP_LIB points benchmarks at an alternative build (e.g. the timing-only ablation library).
LIB_PATH = os.environ.get("FA2_HIP_LIB") or os.path.join(_HERE, "libfa2_hip.so")

FA2_DTYPE_F32, FA2_DTYPE_F16, FA2_DTYPE_BF16, FA2_DTYPE_F8E5M2, FA2_DTYPE_F8E4M3, FA2_DTYPE_F64 = range(6)
VARIANT_AUTO, VARIANT_GENERIC, VARIANT_MFMA16, VARIANT_MFMA16_W8, VARIANT_MFMA32, VARIANT_MFMA16P, \
    VARIANT_MFMA16P_W8, VARIANT_MFMA16X, VARIANT_MFMA16D, VARIANT_MFMA16D_W4, VARIANT_MFMA8, \
    VARIANT_MFMA8_W4, VARIANT_MFMA16S, VARIANT_MFMA16S_W4, VARIANT_MFMA16H, VARIANT_MFMA16H_W4, \
    VARIANT_MFMA8X, VARIANT_MFMA8X_W4, VARIANT_MFMA8U, VARIANT_MFMA16K, \
    VARIANT_MFMA16K_R2K2 = range(21)
VARIANT_MFMA16K_R2K4 = 23  # 21 / 22 are the experimental MFMA16P schedules
VARIANT_A64 = 24
VARIANT_A16 = 25
VARIANT_A8 = 26
VARIANT_A64D = 27
# The variants include/fa2_fwd.h publishes -- what libfa2_hip.so runs.
VARIANTS = {"auto": VARIANT_AUTO, "generic": VARIANT_GENERIC, "mfma16": VARIANT_MFMA16, "mfma16_w8": VARIANT_MFMA16_W8,
            "mfma32": VARIANT_MFMA32, "mfma16d": VARIANT_MFMA16D, "mfma16d_w4": VARIANT_MFMA16D_W4, "mfma16h": VARIANT_MFMA16H,
            "mfma16h_w4": VARIANT_MFMA16H_W4, "mfma8x": VARIANT_MFMA8X, "mfma8x_w4": VARIANT_MFMA8X_W4, "mfma16k": VARIANT_MFMA16K,
            "mfma16k_r2k2": VARIANT_MFMA16K_R2K2, "mfma16k_r2k4": VARIANT_MFMA16K_R2K4, "a64": VARIANT_A64, "a16": VARIANT_A16, "a8": VARIANT_A8, "a64d": VARIANT_A64D}
# Experimental kernels, A/B baselines and timing-only ablations: they exist only in the experiments / ablation builds of the library
# (`make -C flash_attention_dlrs_amd/csrc experiments|abl`, csrc/fa2_experiments.h), which benchmarks/ load through FA2_HIP_LIB --
# the names are published only when such a build is the one loaded.
EXPERIMENTAL_VARIANTS = {
    "mfma16p": VARIANT_MFMA16P, "mfma16p_w8": VARIANT_MFMA16P_W8, "mfma16x": VARIANT_MFMA16X, "mfma16s": VARIANT_MFMA16S,
    "mfma16s_w4": VARIANT_MFMA16S_W4, "mfma8": VARIANT_MFMA8, "mfma8_w4": VARIANT_MFMA8_W4, "mfma8u": VARIANT_MFMA8U,
    "mfma16p_x1": VARIANT_MFMA16P + 16, "mfma16p_w8_x1": VARIANT_MFMA16P_W8 + 16,
    "abl_noexp": VARIANT_MFMA16P_W8 + 32, "abl_nosum": VARIANT_MFMA16P_W8 + 64,
    "abl_nomax": VARIANT_MFMA16P_W8 + 128, "abl_all": VARIANT_MFMA16P_W8 + 224,
    "abl_nobar": VARIANT_MFMA16P_W8 + 256, "abl_noload": VARIANT_MFMA16P_W8 + 512,
    "abl_skeleton": VARIANT_MFMA16P_W8 + 736, "abl_nobar_only": VARIANT_MFMA16P_W8 + 192 * 16,
    "mfma16p_w8_x2": VARIANT_MFMA16P_W8 + 1024, "mfma16p_x2": VARIANT_MFMA16P + 1024,
    "x_noexp": VARIANT_MFMA16X + 2048 * 1, "x_nosoftmax": VARIANT_MFMA16X + 2048 * 3,
    "x_nolds": VARIANT_MFMA16X + 2048 * 4, "x_mfma_only": VARIANT_MFMA16X + 2048 * 7,
    "x_nostage": VARIANT_MFMA16X + 2048 * 8, "x_bare": VARIANT_MFMA16X + 2048 * 15,
    "x_nobar": VARIANT_MFMA16X + 2048 * 16, "x_noload": VARIANT_MFMA16X + 2048 * 32,
    "x_nobar_noload": VARIANT_MFMA16X + 2048 * 48}
if os.path.basename(LIB_PATH) in ("libfa2_hip_exp.so", "libfa2_hip_abl.so"):
    VARIANTS.update(EXPERIMENTAL_VARIANTS)

# Every symbol include/fa2_fwd.h declares (tests/test_abi.py checks the export list against the header).
SYMBOLS = ("fa2_fwd", "fa2_fwd_variant", "fa2_query_tile", "fa2_query_tile_ex", "fa2_query_tile_scaled", "fa2_version", "fa2_last_error")
# ... and include/fa2_bwd.h
BWD_SYMBOLS = ("fa2_bwd", "fa2_bwd_variant")
BWD_VARIANTS = {"auto": 0, "generic": 1, "mfma16": 2, "mfma32": 3}

_lib = None


class Fa2LibraryMissing(ImportError):
    pass


def lib():
    """Load the C-ABI library.  No fallback: a missing build is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Fa2LibraryMissing(
                f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C flash_attention_dlrs_amd/csrc`. There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        i64p = ctypes.POINTER(ctypes.c_int64)
        vp = ctypes.c_void_p
        common = [vp, vp, vp, vp, vp, i64p, i64p, i64p, i64p, i64p] + [ctypes.c_int32] * 6 + [ctypes.c_float, vp]
        l.fa2_fwd.restype = ctypes.c_int
        l.fa2_fwd.argtypes = common
        l.fa2_fwd_variant.restype = ctypes.c_int
        l.fa2_fwd_variant.argtypes = common + [ctypes.c_int32]
        # second prototype of the same entry for the hot path: the five stride pointers as plain addresses into ONE
        # int64 array (fa2_fwd below), which saves building five ctypes arrays per launch
        l.fwd_variant_addr = ctypes.CFUNCTYPE(ctypes.c_int, *([vp] * 10 + [ctypes.c_int32] * 6 + [ctypes.c_float, vp, ctypes.c_int32]))(
            ("fa2_fwd_variant", l))
        l.fa2_query_tile.restype = ctypes.c_int
        l.fa2_query_tile.argtypes = [ctypes.c_int32] * 4 + [ctypes.POINTER(ctypes.c_int32)]
        l.fa2_query_tile_ex.restype = ctypes.c_int
        l.fa2_query_tile_ex.argtypes = [ctypes.c_int32] * 6 + [ctypes.POINTER(ctypes.c_int32)]
        l.fa2_query_tile_scaled.restype = ctypes.c_int
        l.fa2_query_tile_scaled.argtypes = [ctypes.c_int32] * 6 + [ctypes.c_float, ctypes.POINTER(ctypes.c_int32)]
        bwd = [vp] * 10 + [i64p] * 9 + [ctypes.c_int32] * 6 + [ctypes.c_float, vp]
        l.fa2_bwd.restype = ctypes.c_int
        l.fa2_bwd.argtypes = bwd
        l.fa2_bwd_variant.restype = ctypes.c_int
        l.fa2_bwd_variant.argtypes = bwd + [ctypes.c_int32]
        l.bwd_variant_addr = ctypes.CFUNCTYPE(ctypes.c_int, *([vp] * 19 + [ctypes.c_int32] * 6 + [ctypes.c_float, vp, ctypes.c_int32]))(
            ("fa2_bwd_variant", l))
        l.fa2_version.restype = ctypes.c_char_p
        l.fa2_last_error.restype = ctypes.c_char_p
        _lib = l
    return _lib


def version():
    return lib().fa2_version().decode()


def query_tile(N, d, dtype_enum, causal=False, B=None, H=None, scale=None):
    """(variant, B_r, B_c, waves) the static table picks; the choice depends on the grid size, so pass B and H for the
    variant an actual (B, H, N, d) launch takes (default: a large grid, B = 64, H = 8) -- and, for f16, on the softmax scale
    (default 1, the reference's)"""
    out = (ctypes.c_int32 * 4)()
    if scale is not None:
        rc = lib().fa2_query_tile_scaled(64 if B is None else B, 8 if H is None else H, N, d, dtype_enum, int(bool(causal)), float(scale), out)
    elif B is None or H is None:
        rc = lib().fa2_query_tile(N, d, dtype_enum, int(bool(causal)), out)
    else:
        rc = lib().fa2_query_tile_ex(B, H, N, d, dtype_enum, int(bool(causal)), out)
    if rc != 0:
        _raise(rc)
    return tuple(out)


def _raise(rc):
    msg = lib().fa2_last_error().decode()
    if rc == -2:
        raise TypeError(f"fa2_fwd: {msg}")           # reference: TypeError for unsupported dtype (torch.py:18)
    if rc in (-1, -3):
        raise ValueError(f"fa2_fwd: {msg}")          # reference: ValueError for bad shapes (torch.py:28-32)
    raise RuntimeError(f"fa2_fwd rc={rc}: {msg}")


def _i64(vals):
    return (ctypes.c_int64 * len(vals))(*vals)


def _raw_stream(index):
    """Handle of torch's current stream on device `index`."""
    try:
        return torch._C._cuda_getCurrentRawStream(index)
    except AttributeError:  # older / newer torch without the private accessor
        return torch.cuda.current_stream(index).cuda_stream


def fa2_fwd(Q, K, V, O, L, dtype_enum, causal=False, scale=1.0, variant=VARIANT_AUTO):
    """Launch the forward on the current stream of Q's device.  Tensors are (B, H, N, d) with
    arbitrary strides; O (B, H, N, d) and L (B, H, N, 1) are pre-allocated by the caller exactly as
    the reference's host glue does (torch.py:50-51)."""
    if Q.device.type != "cuda":
        raise NotImplementedError("Q, K, V must be on the same CUDA device")
    B, H, N, d = Q.shape
    LB, LH = L.stride(0), L.stride(1)

    def launch():
        # one int64 array for the 18 strides (five ctypes arrays cost ~2 us), raw stream handle without the Stream object
        st = (ctypes.c_int64 * 18)(*Q.stride(), *K.stride(), *V.stride(), *O.stride(), LB, LH)
        base = ctypes.addressof(st)
        return lib().fwd_variant_addr(
            Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), L.data_ptr(),
            base, base + 32, base + 64, base + 96, base + 128,
            B, H, N, d, int(dtype_enum), int(bool(causal)), float(scale), _raw_stream(Q.device.index), int(variant))
    # the library launches on the CURRENT HIP device: switch only if it is not the tensors' one (the device guard
    # costs several microseconds of the ~15 a small launch takes on the host)
    if torch.cuda.current_device() == Q.device.index:
        rc = launch()
    else:
        with torch.cuda.device(Q.device):
            rc = launch()
    if rc != 0:
        _raise(rc)


def fa2_bwd(Q, K, V, O, dO, L, dQ, dK, dV, D, dtype_enum, causal=False, scale=1.0, variant=0):
    """Launch the backward (include/fa2_bwd.h) on the current stream of Q's device: the counterpart of the
    reference's bwd_D_kernel + bwd_kernel launches (torch.py:124-155).  All buffers, the float32 scratch D
    (2, B, H, N, 1) included, are allocated by the caller as the reference's glue does (torch.py:101-105)."""
    if Q.device.type != "cuda":
        raise NotImplementedError("Q, K, V must be on the same CUDA device")
    B, H, N, d = Q.shape
    assert D.is_contiguous() and D.numel() == 2 * B * H * N

    def launch():
        st = (ctypes.c_int64 * 34)(*Q.stride(), *K.stride(), *V.stride(), *O.stride(), *dO.stride(), *dQ.stride(),
                                    *dK.stride(), *dV.stride(), L.stride(0), L.stride(1))
        base = ctypes.addressof(st)
        return lib().bwd_variant_addr(
            Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), dO.data_ptr(), L.data_ptr(),
            dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), D.data_ptr(),
            *(base + 32 * k for k in range(9)),
            B, H, N, d, int(dtype_enum), int(bool(causal)), float(scale), _raw_stream(Q.device.index), int(variant))
    if torch.cuda.current_device() == Q.device.index:
        rc = launch()
    else:
        with torch.cuda.device(Q.device):
            rc = launch()
    if rc != 0:
        _raise(rc)
