"""Run-time tile auto-selection -- counterpart of the reference's Triton autotuner
(src/flash_attention_kernels.py:11-15 + src/autotune_configs.py:24-201: 114 configs, pruned, benchmarked on the
first call of every (B, H, N, d)).

The library ships a static gfx950 table (`fa2_query_tile`, csrc/fa2_api.hip).  This module is the optional
on-box tuner on top of it: with `FA2_AUTOTUNE=1` (or `enable()`), the first forward of a new key times every
kernel variant that supports the problem on the live device and remembers the fastest; the choices persist in a
JSON file so later processes skip the measurement.  Differences from the reference's scheme, on purpose:

  * key = (dtype, d, N bucket, causal): the choice does not depend on B or H (the reference re-tunes per (B, H, N, d));
    N is bucketed to the next power of two, the grid shape only changes which tile fills the 256 CUs, so the
    number of work units B*H*ceil(N/256) is folded in as "small" / "large" (< / >= 512 units).
  * candidates = the handful of hand-written kernel variants (not a tile-parameter sweep): every one of them is
    parity-tested, so tuning can never change results beyond the documented tolerances.
  * timing = HIP events around back-to-back launches on the caller's tensors (inputs are read-only).
"""
import json
import os

import torch

from . import _lib

_ENABLED = os.environ.get("FA2_AUTOTUNE", "0") not in ("", "0")
_PATH = os.environ.get("FA2_TUNE_TABLE") or os.path.join(os.path.expanduser("~"), ".cache", "fa2_hip_tile_table.json")
_table = None

# variants worth timing per dtype family (all covered by tests/test_fwd_parity.py)
_CANDIDATES = {
    "16": ("a64", "a16", "a64d", "mfma16d", "mfma16d_w4", "mfma16h", "mfma16h_w4", "mfma16_w8", "mfma16k", "mfma16k_r2k2", "mfma16k_r2k4"),
    "8": ("mfma8x", "mfma8x_w4", "a8"),
}


def enable(on=True, path=None):
    global _ENABLED, _PATH, _table
    _ENABLED = bool(on)
    if path:
        _PATH, _table = path, None


def enabled():
    return _ENABLED


def _load():
    global _table
    if _table is None:
        try:
            with open(_PATH) as f:
                _table = json.load(f)
        except (OSError, ValueError):
            _table = {}
    return _table


def _save():
    try:
        os.makedirs(os.path.dirname(_PATH), exist_ok=True)
        tmp = _PATH + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(_table, f, indent=1, sort_keys=True)
        os.replace(tmp, _PATH)
    except OSError:
        pass  # a read-only home directory only costs the persistence


def key_of(Q, causal):
    B, H, N, d = Q.shape
    nb = 1 << max(N - 1, 0).bit_length()
    units = B * H * ((N + 255) // 256)
    return f"{str(Q.dtype).split('.')[-1]}:d{d}:N{nb}:{'causal' if causal else 'full'}:{'large' if units >= 512 else 'small'}"


def table_id(device):
    """the persisted table is per device model and library version (a new kernel build re-tunes)"""
    name = torch.cuda.get_device_name(device) if torch.device(device).type == "cuda" else "cpu"
    return f"{name}|{_lib.version()}"


def _time(Q, K, V, O, L, dtype_enum, causal, scale, variant, iters=5):
    for _ in range(2):
        _lib.fa2_fwd(Q, K, V, O, L, dtype_enum, causal=causal, scale=scale, variant=variant)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        _lib.fa2_fwd(Q, K, V, O, L, dtype_enum, causal=causal, scale=scale, variant=variant)
    b.record()
    torch.cuda.synchronize(Q.device)
    return a.elapsed_time(b) / iters


def pick(Q, K, V, O, L, dtype_enum, causal, scale):
    """Variant id for this problem: the tuned choice if tuning is on (measuring it first if the key is new),
    otherwise VARIANT_AUTO (the static table)."""
    if not _ENABLED:
        return _lib.VARIANT_AUTO
    table = _load()
    key = key_of(Q, causal)
    tid = table_id(Q.device)
    if key in table and table[key]["variant"] in _lib.VARIANTS and table[key].get("table_id") == tid:
        return _lib.VARIANTS[table[key]["variant"]]
    fam = "16" if Q.dtype in (torch.float16, torch.bfloat16) else "8" if Q.element_size() == 1 else None
    if fam is None:  # fp32 / fp64: one MFMA kernel and the generic one -- nothing to choose
        return _lib.VARIANT_AUTO
    # clock ramp: an idle MI355X runs its first milliseconds below the sustained clock (bench.py) -- spin first,
    # then time the candidates in two interleaved rounds and keep each one's best
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        _time(Q, K, V, O, L, dtype_enum, causal, scale, _lib.VARIANT_AUTO, iters=10)
    results = {}
    for _ in range(2):
        for name in ("auto",) + _CANDIDATES[fam]:
            try:
                ms = _time(Q, K, V, O, L, dtype_enum, causal, scale, _lib.VARIANTS[name])
            except (TypeError, ValueError, RuntimeError):
                continue  # this variant does not support the problem (d, strides, alignment)
            results[name] = min(ms, results.get(name, ms))
    best = min(results, key=results.get)
    # keep the static choice unless a candidate beats it by more than the run-to-run noise
    if results[best] > 0.98 * results["auto"]:
        best = "auto"
    table[key] = {"variant": best, "ms": {k: round(v, 5) for k, v in results.items()},
                  "device": torch.cuda.get_device_name(Q.device), "table_id": tid}
    _save()
    return _lib.VARIANTS[best]
