"""CPU tests of the generated assembly kernel (variant a64) -- no GPU needed.

The generator's instruction stream is (1) checked for wait-state violations, (2) assembled for gfx950, and (3) executed by the
wave64 emulator (flash_attention_dlrs_amd/csrc/asm/emu.py: test infrastructure, asynchronous loads poisoned until their
s_waitcnt) for small problems, whose O and L are compared with the CPU oracle (oracle/fa2_oracle.c) in the kernel's I/O
dtype.  Tolerances: bf16 |O - oracle| <= 5e-2 (the bar of tests/test_fwd_parity.py), f16 6e-3, L one ulp of the dtype.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from flash_attention_dlrs_amd.csrc.asm import harness
from flash_attention_dlrs_amd.csrc.asm.check import check
from flash_attention_dlrs_amd.csrc.asm.fa2_a64_gen import KARG_SIZE, Gen, module_text

O_TOL = {"bf16": 5e-2, "f16": 6e-3}
ORACLE_DT = {"bf16": "bfloat16", "f16": "float16"}
_PROGS = {}


def prog(dtype, causal, split=True):
    if (dtype, causal, split) not in _PROGS:
        g = Gen(dtype, causal, split=split)
        _PROGS[(dtype, causal, split)] = (g, g.build())
    return _PROGS[(dtype, causal, split)]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("causal", [False, True])
def test_generated_stream_has_no_wait_state_violation(dtype, causal):
    _, p = prog(dtype, causal)
    assert check(p, verbose=False) == []


@pytest.mark.parametrize("kw", [dict(ragged=True, causal=False), dict(ragged=True, causal=True), dict(split=False, causal=True)])
def test_the_other_shipped_and_ab_streams_have_no_wait_state_violation(kw):
    """the ragged kernels ship in the same code object, the contiguous row map is the A/B variant of the experiments build"""
    kw = dict(kw)
    g = Gen("bf16", kw.pop("causal"), **kw)
    assert check(g.build(), verbose=False) == []


def test_store_data_hazard_rule():
    """check.py R10: a VALU write of the data registers of a store wider than 64 bits needs two wait states behind the store"""
    from flash_attention_dlrs_amd.csrc.asm.check import fix
    from flash_attention_dlrs_amd.csrc.asm.isa import I, S, V
    st = I("buffer_store_dwordx4", V(8, 4), V(20), S(60, 4), S(88), offen=1)
    bad = [st, I("v_mov_b32", V(9), 0)]
    assert [e[1] for e in check(bad, verbose=False)] == ["R10 store data overwritten"]
    assert len(check([st, I("s_nop", 0), I("v_mov_b32", V(9), 0)], verbose=False)) == 1      # one wait state is not enough
    assert check([st, I("s_nop", 1), I("v_mov_b32", V(9), 0)], verbose=False) == []
    assert check([st, I("v_mov_b32", V(20), 0)], verbose=False) == []          # the address register is read at issue
    fixed, added = fix(bad)
    assert added == 2 and check(fixed, verbose=False) == []


def test_generated_module_assembles_for_gfx950(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    gens = [prog(dt, c)[0] for dt in ("bf16", "f16") for c in (False, True)]
    src = tmp_path / "a64.s"
    src.write_text(module_text(gens))
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src),
                           "-o", str(tmp_path / "a64.o")])
    assert KARG_SIZE == 192   # the packed struct A64Args of fa2_a64.hip (static_assert there)


def _run(oracle, dtype, causal, B, H, N, scale=1.0, seed=0, spike=False, **kw):
    rng = np.random.default_rng(seed)
    Q, K, V = (rng.standard_normal((B, H, N, 128)).astype(np.float32) for _ in range(3))
    if spike:   # a late jump of one row's maximum far beyond the deferred-rescale threshold (60 / 15.875 log2 units)
        K[:, :, N - 40] = 8.0 * Q[:, :, 5]
    _, p = prog(dtype, causal)
    O, L, _ = harness.run(p, Q, K, V, dtype=dtype, causal=causal, scale=scale, **kw)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    O_ref, L_ref = oracle.forward(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, scale=scale, B_r=64, B_c=64)
    assert not np.isnan(O).any()
    assert np.abs(O - O_ref).max() <= O_TOL[dtype], np.abs(O - O_ref).max()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(L_ref).max())) - (7 if dtype == "bf16" else 10))
    assert np.abs(L - L_ref[..., 0] if L_ref.ndim == 4 else L - L_ref).max() <= 1.01 * ulp


@pytest.mark.parametrize("dtype,causal", [("bf16", False), ("bf16", True), ("f16", True)])
def test_emulated_kernel_matches_oracle_one_job(oracle, dtype, causal):
    _run(oracle, dtype, causal, 1, 1, 256)


def test_emulated_kernel_job_stream_and_wave_order(oracle):
    # three jobs on one workgroup (seam, next-job prefetch, epilogue between jobs), waves released in a permuted order
    _run(oracle, "bf16", True, 1, 3, 256, nwg=1, order=[2, 0, 3, 1], seed=1)


def test_emulated_causal_light_jobs_walk_downwards(oracle):
    """causal with bit 25 of the decode word set (the host's default for N <= 4096): the light job of a unit (query block u) walks
    its non-diagonal key tiles DOWNWARDS -- stream start / step per job, the jump to the diagonal span in the steady loop's last
    trip; two workgroups walk three units each (query blocks 5 + 0, 4 + 1, 3 + 2 of one head), and the result is what the upward
    walk gives within the tolerance (the order of the tiles moves the deferred maximum, not the mathematics)"""
    _run(oracle, "bf16", True, 1, 2, 1536, nwg=2, pairs=True, seed=2)


def test_emulated_kernel_frequent_rescales_across_jobs(oracle):
    # a low deferral threshold makes every rare path (firing, deferred rescale) run in every step, across job seams
    rng = np.random.default_rng(4)
    Q, K, V = (rng.standard_normal((1, 2, 512, 128)).astype(np.float32) for _ in range(3))
    _, p = prog("bf16", True)
    O, L, _ = harness.run(p, Q, K, V, dtype="bf16", causal=True, nwg=1, thr_override=8.0)
    O_ref, _ = harness.reference(Q, K, V, dtype="bf16", causal=True)
    assert not np.isnan(O).any() and np.abs(O - O_ref).max() <= O_TOL["bf16"]


@pytest.mark.parametrize("dtype,thr,split", [("bf16", None, True), ("f16", None, True), ("bf16", 8.0, True), ("bf16", None, False),
                                             ("bf16", 8.0, False)])
def test_emulated_causal_diagonal_with_large_masked_scores(oracle, dtype, thr, split):
    """the lazily masked diagonal tiles: keys a few positions AHEAD of their query carry scores far above every visible one
    (s = 40 .. 160 log2 units).  The row maxima are taken over them; that must neither leak into m (the firing path masks
    exactly and takes the maxima again) nor into P (the packed-P masking), on the waves that sit on the diagonal (patterns D0 /
    D1 of the split row map, the product default) and in the steady loop's last trip (diagonal tile 0 of a longer job).
    split = False: the contiguous row map kept as an A/B variant (waves below the diagonal: running maximum swapped for +inf)."""
    rng = np.random.default_rng(11)
    B, H, N = 1, 2, 768
    Q, K, V = (rng.standard_normal((B, H, N, 128)).astype(np.float32) * 0.5 for _ in range(3))
    for q, ahead, gain in ((5, 3, 2.0), (40, 20, 4.0), (100, 60, 8.0), (300, 1, 6.0), (517, 50, 3.0), (600, 100, 5.0), (767 - 64, 63, 8.0)):
        K[:, :, q + ahead] = gain * Q[:, :, q]        # masked for row q; visible (and large) only for rows >= q + ahead
    _, p = prog(dtype, True, split)
    kw = dict(thr_override=thr) if thr is not None else {}
    O, L, _ = harness.run(p, Q, K, V, dtype=dtype, causal=True, nwg=1, **kw)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    O_ref, L_ref = oracle.forward(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=True, B_r=64, B_c=64)
    assert not np.isnan(O).any() and np.abs(O - O_ref).max() <= O_TOL[dtype], np.abs(O - O_ref).max()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(L_ref).max())) - (7 if dtype == "bf16" else 10))
    assert np.abs(L - L_ref.reshape(L.shape)).max() <= 1.01 * ulp


@pytest.mark.parametrize("dtype,thr", [("bf16", None), ("f16", 6.0)])
def test_split_row_map_is_bit_identical_to_the_contiguous_one(dtype, thr):
    """the split row map (wave w: 32-row blocks w and w + 4; a hidden (tile, block) pair is not computed) only changes which wave
    owns a row: every row sees the same tiles in the same order with the same 32-row rescale groups, so O and L must come out
    bit for bit as with the contiguous map (whose hidden tiles run with +inf as running maximum) -- three jobs per (b, h) incl.
    a job that is nothing but its diagonal, large masked scores next to the diagonal, and (f16, threshold 6) frequent rescales"""
    rng = np.random.default_rng(5)
    B, H, N = 1, 2, 768
    Q, K, V = (rng.standard_normal((B, H, N, 128)).astype(np.float32) * 0.6 for _ in range(3))
    for q, ahead, gain in ((5, 3, 2.0), (100, 60, 8.0), (300, 1, 6.0), (517, 50, 3.0), (600, 100, 5.0)):
        K[:, :, q + ahead] = gain * Q[:, :, q]
    kw = dict(thr_override=thr) if thr is not None else {}
    out = [harness.run(prog(dtype, True, split)[1], Q, K, V, dtype=dtype, causal=True, nwg=1, **kw)[:2] for split in (True, False)]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert not np.isnan(out[0][0]).any()


@pytest.mark.parametrize("dtype,causal,N", [("bf16", False, 300), ("bf16", False, 600), ("f16", False, 744), ("bf16", True, 448), ("f16", True, 520)])
def test_emulated_ragged_kernels(oracle, dtype, causal, N):
    """N not a multiple of 256: the buffers hold exactly N rows (an unchecked access faults in the emulator), two (b, h) on one
    workgroup.  300: the job's last 256 keys hold 44 real ones (tile 0 partial, tiles 1-3 unreal); 600: 88 (tile 1 partial);
    744: 232 (tile 3 partial)"""
    rng = np.random.default_rng(N)
    Q, K, V = (rng.standard_normal((1, 2, N, 128)).astype(np.float32) for _ in range(3))
    g = Gen(dtype, causal, ragged=True)
    O, L, _ = harness.run(g.build(), Q, K, V, dtype=dtype, causal=causal, nwg=1)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    O_ref, L_ref = oracle.forward_deferred(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, G=32, B_c=64,
                                           thr=harness.A64_THR[dtype])
    assert not np.isnan(O).any() and np.abs(O - O_ref).max() <= O_TOL[dtype]
    ulp = 2.0 ** (np.floor(np.log2(np.abs(L_ref).max())) - (7 if dtype == "bf16" else 10))
    assert np.abs(L - L_ref[..., 0]).max() <= 1.01 * ulp


def test_emulated_kernel_rescale_path(oracle):
    _run(oracle, "bf16", False, 1, 1, 512, spike=True, seed=2)


def test_scc_producer_rule_catches_a_broken_carry_chain():
    """check.py R9: an s_addc_u32 must take its carry from the s_add_u32 of its own pair (a filler that writes SCC in between --
    here a DMA set-up's s_add_u32 -- silently drops the carry of a 64-bit address)"""
    from flash_attention_dlrs_amd.csrc.asm.check import check_scc, fix
    from flash_attention_dlrs_amd.csrc.asm.isa import I, M0, S
    good = [I("s_add_u32", S(88), S(88), S(94)), I("s_addc_u32", S(89), S(89), S(95))]
    assert check_scc(good) == []
    bad = [good[0], I("s_add_u32", M0, S(78), 0x8400), good[1]]
    assert [e[1] for e in check_scc(bad)] == ["R9 scc producer"]
    with pytest.raises(RuntimeError, match="SCC consumed from the wrong producer"):
        fix(bad)
    assert check_scc([I("s_cmp_lg_u32", S(80), 0), I("s_cbranch_scc1", "x")]) == []
    assert len(check_scc([I("s_cmp_lg_u32", S(80), 0), I("s_add_u32", S(70), S(70), S(76)), I("s_cbranch_scc1", "x")])) == 1
