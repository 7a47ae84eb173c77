"""CPU tests of the generated assembly kernel at head size 64 (variant a64d, asm/fa2_a64d_gen.py) -- no GPU needed.

As tests/test_asm_emu.py does for a64: wait-state check, assembly for gfx950, and the wave64 emulator against the CPU oracle in the
kernel's I/O dtype (bf16 |O - oracle| <= 5e-2, f16 6e-3, L one ulp of the dtype), non-causal and causal (split row map, light jobs
walking downwards), with the emulator's LDS bank model on the 128-byte-row tile images.
"""
import os
import subprocess

import numpy as np
import pytest

from flash_attention_dlrs_amd.csrc.asm import emu, harness
from flash_attention_dlrs_amd.csrc.asm.check import check
from flash_attention_dlrs_amd.csrc.asm.fa2_a64d_gen import KARG_SIZE, Gen
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
    assert check(prog(dtype, causal, ragged)[1], verbose=False) == []


def test_generated_module_assembles_for_gfx950(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    src = tmp_path / "a64d.s"
    src.write_text(module_text([prog(dt, c)[0] for dt in ("bf16", "f16") for c in (False, True)]))
    subprocess.check_call([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "a64d.o")])
    assert KARG_SIZE == 192


def _run(oracle, dtype, causal, B, H, N, seed=0, spike=False, spikes=(), spread=1.0, **kw):
    rng = np.random.default_rng(seed)
    Q, K, V = (rng.standard_normal((B, H, N, 64)).astype(np.float32) * spread for _ in range(3))
    if spike:   # a late jump of one row's maximum far beyond the deferred-rescale threshold (60 / 15.875 log2 units)
        K[:, :, N - 40] = 12.0 * Q[:, :, 5]
    for q, ahead, gain in spikes:
        K[:, :, q + ahead] = gain * Q[:, :, q]
    O, L, _ = harness.run(prog(dtype, causal, bool(N % 256))[1], Q, K, V, dtype=dtype, causal=causal, **kw)
    rd = lambda x: harness.from_dt(harness.to_dt(x, dtype), dtype)
    if N % 64:      # (the plain restatement wants whole tiles: the deferred-maximum mode takes any N)
        O_ref, L_ref = oracle.forward_deferred(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, G=32, B_c=64, thr=harness.A64_THR[dtype])
    else:
        O_ref, L_ref = oracle.forward(rd(Q), rd(K), rd(V), ORACLE_DT[dtype], causal=causal, B_r=64, B_c=64)
    assert not np.isnan(O).any()
    assert np.abs(O - O_ref).max() <= O_TOL[dtype], np.abs(O - O_ref).max()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(L_ref).max())) - (7 if dtype == "bf16" else 10))
    assert np.abs(L - L_ref.reshape(L.shape)).max() <= 1.01 * ulp


@pytest.mark.parametrize("dtype,causal", [("bf16", False), ("bf16", True), ("f16", True)])
def test_emulated_kernel_matches_oracle_one_job(oracle, dtype, causal):
    _run(oracle, dtype, causal, 1, 1, 256)


def test_emulated_ragged_kernels(oracle):
    """N not a multiple of 256: range-checked descriptors (num_records counts rows of 128 bytes here), the key tail of the job's last
    four tiles masked; in f16, whose tolerance shows a row that is off by one key's weight"""
    _run(oracle, "f16", False, 1, 1, 300, spread=0.6)
    _run(oracle, "f16", False, 1, 1, 257, seed=1, spread=0.6)
    _run(oracle, "bf16", False, 1, 3, 513, nwg=1, seed=2)
    _run(oracle, "f16", True, 1, 2, 700, seed=3, spread=0.6)
    rng = np.random.default_rng(320)
    Q, K, V = (rng.standard_normal((1, 1, 320, 64)).astype(np.float32) * 0.6 for _ in range(3))
    O, _, _ = harness.run(prog("f16", False, True)[1], Q, K, V, dtype="f16", causal=False)
    assert np.abs(O - harness.reference(Q, K, V, dtype="f16", causal=False)[0]).max() <= 1.5e-3


def test_emulated_kernel_job_stream_and_wave_order(oracle):
    _run(oracle, "bf16", True, 1, 3, 256, nwg=1, order=[2, 0, 3, 1], seed=1)
    _run(oracle, "bf16", False, 2, 3, 256, nwg=2, seed=6, pow2=False)


def test_emulated_steady_loop_rescale_path_and_downward_light_jobs(oracle):
    _run(oracle, "bf16", False, 1, 1, 1024, seed=3)
    _run(oracle, "bf16", False, 1, 1, 512, spike=True, seed=2)
    _run(oracle, "bf16", True, 1, 2, 512, nwg=1, thr_override=8.0, seed=4)
    _run(oracle, "bf16", True, 1, 2, 1536, nwg=2, pairs=True, seed=2)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_emulated_causal_diagonal_with_large_masked_scores(oracle, dtype):
    sp = ((5, 3, 4.0), (40, 20, 8.0), (100, 60, 16.0), (300, 1, 12.0), (517, 50, 6.0), (600, 100, 10.0), (767 - 64, 63, 16.0))
    _run(oracle, dtype, True, 1, 2, 768, nwg=1, seed=11, spread=0.5, spikes=sp)


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
    Q, K, V = (rng.standard_normal((1, 1, 512, 64)).astype(np.float32) for _ in range(3))
    harness.run(prog("bf16", False)[1], Q, K, V, dtype="bf16", causal=False)
    for tag in ("kread", "vread", "qread"):
        assert stats[tag][0] > 0 and stats[tag][1] == 0, (tag, stats[tag])
