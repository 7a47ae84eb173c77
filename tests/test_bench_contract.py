"""bench.py prints ONE JSON line that carries the driver's contract fields, the roofline and cpu_baseline
objects; also exercised through torch.distributed.run (world_size 1) so the N > 1 code path (process group,
barrier, max-over-ranks, sharded gather) at least executes on one GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_single_process_json_contract():
    j = _run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2"])
    assert REQUIRED <= set(j)
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["higher_is_better"] is True
    assert j["unit"] == "TFLOP/s" and j["dtype"] == "bf16" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert j["value"] > 50 * c["value"]  # sanity: the GPU path is not the CPU path


def test_distributed_launcher_world1():
    j = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
              "--master-addr", "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "1", "--steps", "3",
              "--warmup", "1", "--no-cpu-baseline", "--gather"])
    assert j["n_gpus"] == 1 and j["value"] > 0
    # launched by torch.distributed.run the bench initialises the nccl (= RCCL) group even at world size 1 and --gather puts
    # the all-gather of O into the step
    g = j["extras"]["gather"]
    assert g["included_in_step"] is True and g["world_size"] == 1 and g["ms_compute_plus_gather"] > 0


def test_gpus_flag_starts_its_own_rank_processes():
    """`python bench.py --gpus N` as a plain command (no torch.distributed.run around it): the parent spawns the ranks, relays
    rank 0's line and exit code.  One GPU here, so the branch is forced with --self-launch at N = 1."""
    j = _run([sys.executable, "bench.py", "--gpus", "1", "--self-launch", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert j["n_gpus"] == 1 and j["value"] > 0
    assert j["n_ranks_seen_by_rccl"] == 1        # the rank process ran under torch.distributed.run with the nccl group up


def test_query_tile_reports_the_variant_the_launch_takes():
    from flash_attention_dlrs_amd import _lib
    # c2 (B2 H8 N1024 d64 fp16) is a small grid: the key-split kernel, not the large-grid answer
    assert _lib.query_tile(1024, 64, _lib.FA2_DTYPE_F16, False, B=2, H=8)[0] == _lib.VARIANT_MFMA16K_R2K4
    assert _lib.query_tile(4096, 128, _lib.FA2_DTYPE_BF16, True, B=4, H=32)[0] == _lib.VARIANT_A64
    j = _run([sys.executable, "bench.py", "--config", "c2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert j["config"]["tile"]["variant"] == _lib.VARIANT_MFMA16K_R2K4 and "N=1024" in j["metric"]


@pytest.mark.parametrize("mode", ["fwd", "bwd"])
def test_sweep_writes_the_csv_the_reference_plot_script_reads(tmp_path, mode):
    """benchmarks/bench_sweep.py = the reference's src/bench.py: the file name and schema src/plot_bench_results.py:41-57,
    102-126 consumes -- an `N` column and one column of mean milliseconds per provider display name, among them the
    reference's own `Torch Math [FLOAT16]` (src/bench.py:38-41)."""
    import pandas as pd
    out = subprocess.run([sys.executable, "benchmarks/bench_sweep.py", "--n-max-log", "9", "--mode", mode, "--out-dir", str(tmp_path)],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    df = pd.read_csv(tmp_path / f"fused-attention-B8-H16-d128-{mode}-float16.csv", sep=",")   # plot_bench_results.py:53-57
    df["N"] = df["N"].astype("int32")                                                           # plot_bench_results.py:73
    assert list(df["N"]) == [128, 256, 512]
    cols = [c for c in df.columns if c != "N"]
    assert "MI355X HIP FA-2 [FLOAT16]" in cols and "Torch Math [FLOAT16]" in cols and "Torch FA-2 [FLOAT16]" in cols
    assert (df["MI355X HIP FA-2 [FLOAT16]"] > 0).all() and (df["Torch Math [FLOAT16]"] > 0).all()
    tf = pd.read_csv(tmp_path / f"fused-attention-B8-H16-d128-{mode}-float16-tflops.csv")
    assert list(tf.columns) == list(df.columns) and (tf["MI355X HIP FA-2 [FLOAT16]"] > 0).all()
