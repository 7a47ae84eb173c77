"""The C-ABI library loads and exports every symbol include/fa2_fwd.h declares; argument validation
(which happens before any HIP call) returns the documented codes.  No compute, no GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT

from flash_attention_dlrs_amd import _lib

HEADER = os.path.join(ROOT, "include", "fa2_fwd.h")
HEADER_BWD = os.path.join(ROOT, "include", "fa2_bwd.h")


def _declared(header):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fa2_\w+)\s*\(", src)))


def declared_functions():
    return sorted(_declared(HEADER) + _declared(HEADER_BWD))


def test_header_and_binding_agree():
    assert _declared(HEADER) == sorted(_lib.SYMBOLS)
    assert _declared(HEADER_BWD) == sorted(_lib.BWD_SYMBOLS)


def test_library_exports_every_declared_symbol():
    l = _lib.lib()
    for name in declared_functions():
        assert getattr(l, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in declared_functions():
        assert re.search(rf"\bT {name}\b", out), name


def test_exported_symbols_are_only_the_c_abi():
    """EVERY defined symbol of the dynamic table, whatever its type or mangling (the library links with the version script
    csrc/fa2_exports.map: C++ launchers, helpers and hipcc's per-unit markers are local)"""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == set(declared_functions()), sorted(exported ^ set(declared_functions()))


def test_header_compiles_as_plain_c(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "fa2_fwd.h"\n#include "fa2_bwd.h"\n'
                 'int main(void){return FA2_OK + FA2_BWD_VARIANT_AUTO + (fa2_version()!=0);}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                           str(c), "-o", str(tmp_path / "t.o")])


def test_version_string():
    assert _lib.version().startswith("fa2-hip ") and "gfx950" in _lib.version()


def _call(N=64, d=64, dtype=_lib.FA2_DTYPE_F32, ptr=0x1000, B=1, H=1, strides=None, scale=1.0):
    s = strides or (H * N * d, N * d, d, 1)
    i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
    return _lib.lib().fa2_fwd(ptr, ptr, ptr, ptr, ptr, i64(s), i64(s), i64(s), i64(s), i64((H * N, N)),
                              B, H, N, d, dtype, 0, scale, None)


@pytest.mark.parametrize("kwargs,code,needle", [
    (dict(ptr=0), -1, "null"),
    (dict(B=0), -1, "positive"),
    (dict(N=0), -3, "N must be"),
    (dict(d=0), -1, "positive"),
    (dict(d=1024), -2, "[1, 512]"),
    (dict(dtype=99), -2, "dtype"),
    (dict(strides=(-1, 1, 1, 1)), -1, "negative"),
    (dict(scale=float("nan")), -1, "NaN"),
])
def test_validation_codes_before_any_launch(kwargs, code, needle):
    assert _call(**kwargs) == code
    assert needle in _lib.lib().fa2_last_error().decode()


def _call_bwd(N=64, d=64, dtype=_lib.FA2_DTYPE_F32, ptr=0x1000, B=1, H=1, dptr=0x1000):
    s = (H * N * d, N * d, d, 1)
    i64 = lambda v: (ctypes.c_int64 * len(v))(*v)
    st = [i64(s) for _ in range(8)]
    return _lib.lib().fa2_bwd(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, dptr, *st, i64((H * N, N)),
                              B, H, N, d, dtype, 0, 1.0, None)


@pytest.mark.parametrize("kwargs,code,needle", [
    (dict(ptr=0), -1, "null"),
    (dict(dptr=0), -1, "null"),
    (dict(B=0), -1, "positive"),
    (dict(N=0), -3, "N must be"),
    (dict(d=48), -2, "power of two"),
    (dict(dtype=_lib.FA2_DTYPE_F8E5M2), -2, "not supported"),
    (dict(dtype=99), -2, "not supported"),
])
def test_backward_validation_codes_before_any_launch(kwargs, code, needle):
    assert _call_bwd(**kwargs) == code
    assert needle in _lib.lib().fa2_last_error().decode()


def test_exception_mapping_mirrors_reference():
    # reference: TypeError for an unsupported dtype (torch.py:18), ValueError for bad shapes (torch.py:28-32)
    _call(dtype=99)
    with pytest.raises(TypeError):
        _lib._raise(-2)
    with pytest.raises(ValueError):
        _lib._raise(-1)
    with pytest.raises(ValueError):
        _lib._raise(-3)
    with pytest.raises(RuntimeError):
        _lib._raise(-4)


def test_static_tile_table():
    q = _lib.query_tile
    # north-star config c3 and friends take the MFMA paths; odd head sizes and exotic dtypes fall back
    assert q(4096, 128, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64          # north-star shape (c3): the assembly kernel
    assert q(4096, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A16         # long non-causal bf16: the same kernel on 16x16x32
    assert q(2048, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A64
    assert q(4096 + 64, 128, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64      # (N not a multiple of 256: its ragged form)
    assert q(8192 + 64, 128, _lib.FA2_DTYPE_BF16, False)[0] == _lib.VARIANT_A16     # (... and a16's on long jobs)
    assert q(200, 128, _lib.FA2_DTYPE_BF16, True)[0] != _lib.VARIANT_A64            # below one 256-row job
    assert q(4096, 64, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64D           # head size 64: the generated kernel at d = 64
    assert q(4096 + 64, 64, _lib.FA2_DTYPE_BF16, True)[0] == _lib.VARIANT_A64D      # (its ragged form)
    assert q(1024, 64, _lib.FA2_DTYPE_F16)[0] == _lib.VARIANT_A64D
    assert q(256, 128, _lib.FA2_DTYPE_F32)[0] == _lib.VARIANT_MFMA32
    assert q(128, 32, _lib.FA2_DTYPE_F32)[0] == _lib.VARIANT_MFMA32   # (d < 64: the d = 64 kernel with the missing columns zero-filled)
    assert q(128, 64, _lib.FA2_DTYPE_F64)[0] == _lib.VARIANT_GENERIC
    assert q(128, 128, _lib.FA2_DTYPE_F8E5M2)[0] in (_lib.VARIANT_MFMA8X, _lib.VARIANT_MFMA8X_W4)
    assert q(128, 64, _lib.FA2_DTYPE_F8E4M3)[0] == _lib.VARIANT_GENERIC
    # supported N domain is a superset of the reference's (multiples of 16, autotune_configs.py:176-187)
    for N in (16, 48, 100, 4096):
        assert q(N, 64, _lib.FA2_DTYPE_F32)[1] > 0
    # head sizes that are not powers of two stay on the matrix cores when they are multiples of 8 (16-bit) / 4 (fp32): the
    # kernels zero-fill the missing columns on load; anything else runs on the generic kernel
    assert q(128, 40, _lib.FA2_DTYPE_F32)[0] == _lib.VARIANT_MFMA32
    assert q(128, 8, _lib.FA2_DTYPE_F32)[0] == _lib.VARIANT_MFMA32
    assert q(128, 96, _lib.FA2_DTYPE_BF16)[0] in (_lib.VARIANT_MFMA16, _lib.VARIANT_MFMA16_W8)
    assert q(128, 36, _lib.FA2_DTYPE_F16)[0] == _lib.VARIANT_GENERIC
    assert q(128, 37, _lib.FA2_DTYPE_F32)[0] == _lib.VARIANT_GENERIC
    with pytest.raises(TypeError):
        q(128, 513, _lib.FA2_DTYPE_F32)
    # f16 depends on the softmax scale (fa2_query_tile_scaled): at the reference's scale of 1 it rescales every few key tiles, which a
    # one-wave-per-SIMD kernel pays for on half-filled grids and the 16x16x32 form pays twice; at the usual scales it follows bf16
    F16, BF16 = _lib.FA2_DTYPE_F16, _lib.FA2_DTYPE_BF16
    assert q(1024, 128, F16, False, B=1, H=16)[0] == _lib.VARIANT_MFMA16K_R2K2 and q(1024, 128, BF16, False, B=1, H=16)[0] == _lib.VARIANT_A64
    assert q(1024, 128, F16, False, B=1, H=16, scale=0.25)[0] == _lib.VARIANT_A64
    assert q(4096, 128, F16, False, B=1, H=64)[0] == _lib.VARIANT_A64 and q(4096, 128, F16, False, B=1, H=64, scale=128 ** -0.5)[0] == _lib.VARIANT_A16
    assert q(4096, 128, BF16, False, B=1, H=8)[0] == _lib.VARIANT_A64          # (a16 only from 192 jobs per 256 CUs on)
    assert q(4096, 64, F16, True, B=1, H=16)[0] != _lib.VARIANT_A64D and q(4096, 64, F16, True, B=1, H=16, scale=0.25)[0] == _lib.VARIANT_A64D
    assert q(4096, 64, BF16, True, B=1, H=8)[0] == _lib.VARIANT_MFMA16K_R2K4     # (two 64-row workgroups per CU)


def test_host_side_and_oracle_under_address_sanitizer():
    """`make asan` (csrc/ and oracle/): the host side of the C-ABI library -- validation, strides, tile table, launchers' argument
    packing -- and the oracle built with AddressSanitizer + UBSan, run under this file's CPU tests and the oracle's
    (scripts/run_asan.sh).  CPU build only: GPU sanitizers are not available on the pool."""
    import glob
    if os.environ.get("FA2_HIP_LIB", "").endswith("_asan.so"):
        pytest.skip("already inside the sanitizer run")
    if not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"):
        pytest.skip("no clang AddressSanitizer run time here")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",)}
    out = subprocess.run(["bash", os.path.join(ROOT, "scripts", "run_asan.sh")], capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert " passed" in out.stdout and "failed" not in out.stdout
