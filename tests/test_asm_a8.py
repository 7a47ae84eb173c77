"""CPU tests of the generated fp8 kernel (variant a8, asm/fa2_a8_gen.py: the a64 structure on v_mfma_f32_32x32x64_f8f6f4) -- no GPU.

Wait-state check, assembly for gfx950, and the emulator (fp8 MFMAs, v_cvt_pk_fp8_f32 / _bf8_, ds_read_b64_tr_b8 with the lane map
measured in round 1, the block-scaled MFMA as probed in round 3) against the CPU oracle IN ITS DEFERRED-MAXIMUM MODE
(oracle.forward_deferred(G=32, B_c=64, thr, rounded row sums; ceil_m for the block-scaled kernel, whose running maximum is an
integer): the liberties fa2_mfma8x.hip takes and this kernel shares -- element by element: >= 99 % of O and 97 % of L bit-identical,
the rest within one fp8 step plus half a step of P times max |V| (the bar of tests/test_fwd_parity.py for fa2_mfma8x).
"""
import os
import subprocess

import numpy as np
import pytest

from flash_attention_dlrs_amd.csrc.asm import emu, harness
from flash_attention_dlrs_amd.csrc.asm.check import check
from flash_attention_dlrs_amd.csrc.asm.fa2_a8_gen import KARG_SIZE, Gen
from flash_attention_dlrs_amd.csrc.asm.fa2_a64_gen import module_text

ORACLE_DT = {"e4m3": "float8_e4m3fn", "e5m2": "float8_e5m2"}
_PROGS = {}


def prog(dtype, scaled=True, causal=False, ragged=False):
    if (dtype, scaled, causal, ragged) not in _PROGS:
        g = Gen(dtype, causal, scaled=scaled, ragged=ragged)
        _PROGS[dtype, scaled, causal, ragged] = (g, g.build())
    return _PROGS[dtype, scaled, causal, ragged]


@pytest.mark.parametrize("dtype,causal,ragged", [(dt, c, False) for dt in ["e4m3", "e5m2"] for c in (False, True)] +
                         [("e4m3", False, True), ("e4m3", True, True)])     # (the ragged streams differ in addressing, not by dtype)
def test_generated_stream_has_no_wait_state_violation(dtype, causal, ragged):
    assert check(prog(dtype, causal=causal, ragged=ragged)[1], verbose=False) == []


def test_generated_module_assembles_for_gfx950(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    src = tmp_path / "a8.s"
    src.write_text(module_text([prog(dt, causal=c)[0] for dt in ("e4m3", "e5m2") for c in (False, True)]))
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "a8.o")])
    assert KARG_SIZE == 192


def _run(oracle, dtype, B, H, N, seed=0, spread=0.5, spike=False, scaled=True, causal=False, **kw):
    rng = np.random.default_rng(seed)
    Q, K, V = (rng.standard_normal((B, H, N, 128)).astype(np.float32) * spread for _ in range(3))
    if spike:   # one row's maximum jumps far beyond the deferral threshold in the last tile (spike = the factor on the row's own q)
        K[:, :, N - 40] = float(spike) * Q[:, :, 5]
    O, L, _ = harness.run(prog(dtype, scaled, causal, bool(N % 256))[1], Q, K, V, dtype=dtype, causal=causal, **kw)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    kw.pop("pairs", None)
    O_ref, L_ref = oracle.forward_deferred(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, G=32, B_c=64,
                                           thr=kw.get("thr_override", harness.A64_THR[dtype]), sum_rounded=True, ceil_m=scaled)
    L_ref = L_ref.reshape(L.shape)
    assert not np.isnan(O).any()
    step = 0.25 if dtype == "e5m2" else 0.125
    assert (O == O_ref).mean() >= 0.99 and (L == L_ref).mean() >= 0.97
    assert (np.abs(O - O_ref) <= step * np.abs(O_ref) + 0.5 * step * np.abs(rd(V)).max()).all()
    assert (np.abs(L - L_ref) <= step * np.abs(L_ref) + 1e-3).all()


@pytest.mark.parametrize("dtype", ["e4m3", "e5m2"])
def test_emulated_kernel_matches_oracle_one_job(oracle, dtype):
    _run(oracle, dtype, 1, 1, 256)


def test_emulated_kernel_job_stream_and_wave_order(oracle):
    _run(oracle, "e4m3", 1, 3, 256, nwg=1, order=[2, 0, 3, 1], seed=1)
    _run(oracle, "e4m3", 2, 3, 256, nwg=2, seed=6, pow2=False)


def test_emulated_steady_loop_and_rescale_path(oracle):
    _run(oracle, "e4m3", 1, 1, 1024, seed=3)
    _run(oracle, "e4m3", 1, 1, 512, spike=2.0, seed=2)
    _run(oracle, "e5m2", 1, 2, 512, nwg=1, seed=4, spread=1.0)        # (scores of sigma 16 log2 units: the maximum moves in most steps)
    _run(oracle, "e4m3", 1, 1, 512, seed=5, spread=1.0, thr_override=6.0)


def test_emulated_causal_kernel(oracle):
    """the causal form (split row map, lazily masked diagonal tiles: byte masks on the packed fp8 P, the exact-maximum check at the
    head of the firing path), one job, several jobs per workgroup, light jobs walking downwards"""
    _run(oracle, "e4m3", 1, 1, 256, causal=True)
    _run(oracle, "e4m3", 1, 3, 768, causal=True, nwg=1, seed=3, spread=0.7)
    _run(oracle, "e5m2", 1, 1, 1024, causal=True, seed=2, spread=1.0, pairs=True)


def test_emulated_ragged_kernels(oracle):
    """N not a multiple of 256: range-checked descriptors (rows of 128 bytes), the key tail of the job's last four tiles masked on the
    fp32 scores, wholly unreal tiles through +inf as running maximum; byte stores of L past N dropped"""
    _run(oracle, "e4m3", 1, 1, 300)
    _run(oracle, "e4m3", 1, 1, 257, seed=1)
    _run(oracle, "e5m2", 1, 3, 513, nwg=1, seed=2)
    _run(oracle, "e4m3", 1, 2, 700, causal=True, seed=3, spread=0.7)


def test_emulated_guard_path_rebases_the_accumulators(oracle):
    """m - m_O beyond KMAX = 64 log2 units: the spike row's score is ~ 8 |q|^2 = several hundred"""
    _run(oracle, "e4m3", 1, 1, 512, spike=8.0, seed=2)


def test_emulated_kernel_without_the_block_scale(oracle):
    """the A/B variant of the experiments build: O is rescaled whenever the running maximum moves (round 3's first form)"""
    _run(oracle, "e4m3", 1, 1, 512, spike=2.0, seed=2, scaled=False)
    _run(oracle, "e4m3", 1, 1, 256, seed=7, spread=1.0, scaled=False, thr_override=6.0)


def test_tile_images_are_conflict_free_for_operand_reads(monkeypatch):
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
    Q, K, V = (rng.standard_normal((1, 1, 512, 128)).astype(np.float32) * 0.5 for _ in range(3))
    harness.run(prog("e4m3")[1], Q, K, V, dtype="e4m3", causal=False)
    for tag in ("kread", "vread", "qread"):
        assert stats[tag][0] > 0 and stats[tag][1] == 0, (tag, stats[tag])
