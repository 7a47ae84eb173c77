"""CPU tests of the generated assembly kernel on v_mfma_f32_16x16x32 (variant a16, asm/fa2_a16_gen.py) -- no GPU needed.

As tests/test_asm_emu.py does for a64: the instruction stream is checked for wait-state violations, assembled for gfx950 and
executed by the wave64 emulator against the CPU oracle in the kernel's I/O dtype; in addition the emulator's LDS bank model
(MI355X_MICROARCH.md, section LDS) must find the tile image conflict-free for the K / Q row reads and the transposing V reads.
Tolerances: bf16 |O - oracle| <= 5e-2, f16 6e-3, L one ulp of the dtype.
"""
import os
import subprocess

import numpy as np
import pytest

from flash_attention_dlrs_amd.csrc.asm import emu, harness
from flash_attention_dlrs_amd.csrc.asm.check import check
from flash_attention_dlrs_amd.csrc.asm.fa2_a16_gen import KARG_SIZE, Gen
from flash_attention_dlrs_amd.csrc.asm.fa2_a64_gen import module_text

O_TOL = {"bf16": 5e-2, "f16": 6e-3}
ORACLE_DT = {"bf16": "bfloat16", "f16": "float16"}
_PROGS = {}


def prog(dtype, causal, ragged=False):
    if (dtype, causal, ragged) not in _PROGS:
        g = Gen(dtype, causal, ragged=ragged)
        _PROGS[(dtype, causal, ragged)] = (g, g.build())
    return _PROGS[(dtype, causal, ragged)]


@pytest.mark.parametrize("dtype,causal,ragged", [(dt, c, False) for dt in ["bf16", "f16"] for c in (False, True)] +
                         [("bf16", False, True), ("bf16", True, True)])     # (the ragged streams differ in addressing, not by dtype)
def test_generated_stream_has_no_wait_state_violation(dtype, causal, ragged):
    _, p = prog(dtype, causal, ragged)
    assert check(p, verbose=False) == []


def test_generated_module_assembles_for_gfx950(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    src = tmp_path / "a16.s"
    src.write_text(module_text([prog(dt, c)[0] for dt in ("bf16", "f16") for c in (False, True)]))
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "a16.o")])
    assert KARG_SIZE == 192   # the argument block of fa2_a64.hip serves both shapes


def _run(oracle, dtype, causal, B, H, N, scale=1.0, seed=0, spike=False, spikes=(), spread=1.0, **kw):
    rng = np.random.default_rng(seed)
    Q, K, V = (rng.standard_normal((B, H, N, 128)).astype(np.float32) * spread for _ in range(3))
    if spike:   # a late jump of one row's maximum far beyond the deferred-rescale threshold (60 / 15.875 log2 units)
        K[:, :, N - 40] = 8.0 * Q[:, :, 5]
    for q, ahead, gain in spikes:
        K[:, :, q + ahead] = gain * Q[:, :, q]
    _, p = prog(dtype, causal, ragged=bool(N % 256))
    O, L, _ = harness.run(p, Q, K, V, dtype=dtype, causal=causal, scale=scale, **kw)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    if N % 64:      # (the plain restatement wants whole tiles: the deferred-maximum mode takes any N)
        O_ref, L_ref = oracle.forward_deferred(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, scale=scale, G=32, B_c=64,
                                               thr=harness.A64_THR[dtype])
    else:
        O_ref, L_ref = oracle.forward(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, scale=scale, B_r=64, B_c=64)
    assert not np.isnan(O).any()
    assert np.abs(O - O_ref).max() <= O_TOL[dtype], np.abs(O - O_ref).max()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(L_ref).max())) - (7 if dtype == "bf16" else 10))
    assert np.abs(L - L_ref.reshape(L.shape)).max() <= 1.01 * ulp


@pytest.mark.parametrize("dtype,causal", [("bf16", False), ("bf16", True), ("f16", True)])
def test_emulated_kernel_matches_oracle_one_job(oracle, dtype, causal):
    _run(oracle, dtype, causal, 1, 1, 256)


def test_emulated_ragged_kernels(oracle):
    """N not a multiple of 256 (range-checked descriptors, the key tail masked in the 16x16 score layout: register 8 q' + 4 k' + rr of
    a group <-> key 16 k' + 4 (lane >> 4) + rr), one job and several per workgroup"""
    _run(oracle, "bf16", False, 1, 1, 300)
    _run(oracle, "bf16", False, 1, 2, 513, nwg=1, seed=2)
    for N in (257, 320):      # (one real key / one real tile in the job's last 256: the wholly unreal tiles swap +inf in for the running
        #                        maximum of all FOUR 16-row blocks -- in f16, whose tolerance shows a row that missed it)
        rng = np.random.default_rng(N)
        Q, K, V = (rng.standard_normal((1, 1, N, 128)).astype(np.float32) * 0.6 for _ in range(3))
        O, L, _ = harness.run(prog("f16", False, True)[1], Q, K, V, dtype="f16", causal=False)
        O_ref, _ = harness.reference(Q, K, V, dtype="f16", causal=False)
        assert np.abs(O - O_ref).max() <= 1.5e-3, (N, np.abs(O - O_ref).max())
    _run(oracle, "bf16", True, 1, 1, 448, seed=1)


def test_emulated_kernel_job_stream_and_wave_order(oracle):
    # three jobs on one workgroup (seam, next-job prefetch, epilogue between jobs), waves released in a permuted order
    _run(oracle, "bf16", True, 1, 3, 256, nwg=1, order=[2, 0, 3, 1], seed=1)
    _run(oracle, "bf16", False, 2, 3, 256, nwg=2, seed=6, pow2=False)


def test_emulated_steady_loop_and_rescale_path(oracle):
    _run(oracle, "bf16", False, 1, 1, 1024, seed=3)
    _run(oracle, "bf16", False, 1, 1, 512, spike=True, seed=2)


@pytest.mark.parametrize("dtype,thr,causal", [("bf16", 8.0, True), ("f16", 6.0, False)])
def test_emulated_kernel_frequent_rescales_across_jobs(oracle, dtype, thr, causal):
    # a low deferral threshold makes every rare path (firing with its lane exchanges, deferred rescale) run in every step
    _run(oracle, dtype, causal, 1, 2, 512, nwg=1, thr_override=thr, seed=4)


@pytest.mark.parametrize("dtype,thr", [("bf16", None), ("f16", None), ("bf16", 8.0)])
def test_emulated_causal_diagonal_with_large_masked_scores(oracle, dtype, thr):
    """keys a few positions AHEAD of their query carry scores far above every visible one (40 .. 160 log2 units): the lazily masked
    diagonal tiles must keep them out of m (the firing path masks exactly) and out of P (the packed-P masking), patterns D0 / D1"""
    sp = ((5, 3, 2.0), (40, 20, 4.0), (100, 60, 8.0), (300, 1, 6.0), (517, 50, 3.0), (600, 100, 5.0), (767 - 64, 63, 8.0))
    kw = dict(thr_override=thr) if thr is not None else {}
    _run(oracle, dtype, True, 1, 2, 768, nwg=1, seed=11, spread=0.5, spikes=sp, **kw)


def test_tile_image_is_conflict_free_for_operand_reads(monkeypatch):
    """the K / Q row reads (ds_read_b128 of the 16x16x32 A / B operands) and the transposing V reads (ds_read_b64_tr_b16) of one
    emulated job on the LDS bank model: zero extra cycles"""
    stats = {}
    orig = emu.Workgroup.run

    def run(self, *a, **kw):
        out = orig(self, *a, **kw)
        for k, (n, x) in getattr(self, "lds_conflicts", {}).items():
            n0, x0 = stats.get(k, (0, 0))
            stats[k] = (n0 + n, x0 + x)
        return out
    monkeypatch.setattr(emu.Workgroup, "run", run)
    rng = np.random.default_rng(0)
    Q, K, V = (rng.standard_normal((1, 1, 512, 128)).astype(np.float32) for _ in range(3))
    harness.run(prog("bf16", False)[1], Q, K, V, dtype="bf16", causal=False)
    assert stats["kread"][0] > 0 and stats["vread"][0] > 0 and stats["qread"][0] > 0
    for tag in ("kread", "vread", "qread"):
        assert stats[tag][1] == 0, (tag, stats[tag])
    # (the epilogue's row read-back at a row stride of 272 bytes: one 2-way meeting per lane group, as in a64)
    assert stats["ds_read_b128"][1] <= 4 * stats["ds_read_b128"][0]
