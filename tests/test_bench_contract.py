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
