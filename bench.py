#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FA-2 forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3] [--variant auto] [--gather]

One "step" = one forward pass of the hot path (flash_attention_forward -> C ABI -> HIP kernel) over one
batch of synthetic input that is already resident in HBM.  The workload is BASELINE.json's metric
config c3: B=4 H=32 N=4096 d=128 bf16, causal, scale=1 (the reference's math, src/bench.py:85), inputs
N(0,1) drawn with seed 42 (src/bench.py:26,64-66).  FLOPs follow the convention the reference vendors
(src/flash_attention_openai_tutorial.py:630-633): 4*B*H*N^2*d, halved for causal.

N > 1 (launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL): the path shards over
(batch, head) with NO data-path collective -- every rank runs the c3 workload as its own head shard of a
G-times-wider problem (weak scaling); value = total FLOPs of all ranks / max-over-ranks time.  The
optional exchange step of the north star, an RCCL all-gather of the output shards over xGMI, is timed
separately after the main region and reported under "gather" (use --gather to put it inside the step).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json "configs"
    "c2": dict(B=2, H=8, N=1024, d=64, dtype="fp16", causal=False),
    "c3": dict(B=4, H=32, N=4096, d=128, dtype="bf16", causal=True),
    "c3_noncausal": dict(B=4, H=32, N=4096, d=128, dtype="bf16", causal=False),
    "c4_per_gpu": dict(B=8, H=8, N=8192, d=128, dtype="bf16", causal=False),
    "ref_test": dict(B=32, H=32, N=256, d=128, dtype="f32", causal=False),  # src/test_correctness.py:9-14
    "f32_long": dict(B=2, H=16, N=4096, d=128, dtype="f32", causal=False),
    "ref_bench": dict(B=8, H=16, N=4096, d=128, dtype="fp16", causal=False),  # src/bench.py:8-12 at N=4096
    "ref_bench_bf16": dict(B=8, H=16, N=4096, d=128, dtype="bf16", causal=False),
    "ragged_8k": dict(B=8, H=8, N=8100, d=128, dtype="bf16", causal=False),
    "d64_ragged": dict(B=8, H=16, N=4000, d=64, dtype="bf16", causal=False),
    "d64_ragged_causal": dict(B=8, H=16, N=4000, d=64, dtype="bf16", causal=True),
    "ragged_4k": dict(B=4, H=32, N=4000, d=128, dtype="bf16", causal=False),
    "ragged_8k_causal": dict(B=8, H=8, N=8100, d=128, dtype="bf16", causal=True),
    "c3_fp16": dict(B=4, H=32, N=4096, d=128, dtype="fp16", causal=True),
    "causal_2k_fp16": dict(B=8, H=32, N=2048, d=128, dtype="fp16", causal=True),
    "d64_long": dict(B=8, H=16, N=4096, d=64, dtype="fp16", causal=False),
    "d64_long_causal": dict(B=8, H=16, N=4096, d=64, dtype="bf16", causal=True),
    "d64_8k": dict(B=4, H=16, N=8192, d=64, dtype="bf16", causal=False),
    "d64_8k_causal": dict(B=4, H=16, N=8192, d=64, dtype="bf16", causal=True),
    "d64_2k": dict(B=16, H=32, N=2048, d=64, dtype="bf16", causal=False),
    "d64_2k_causal": dict(B=16, H=32, N=2048, d=64, dtype="bf16", causal=True),
    "n1024": dict(B=8, H=32, N=1024, d=128, dtype="bf16", causal=False),
    "causal_16k": dict(B=1, H=32, N=16384, d=128, dtype="bf16", causal=True),
    "causal_8k": dict(B=2, H=32, N=8192, d=128, dtype="bf16", causal=True),
    "causal_2k": dict(B=8, H=32, N=2048, d=128, dtype="bf16", causal=True),
    "ragged_4000": dict(B=4, H=32, N=4000, d=128, dtype="bf16", causal=False),          # N not a multiple of 256: the ragged a64 kernels
    "ragged_4000_causal": dict(B=4, H=32, N=4000, d=128, dtype="bf16", causal=True),
    "n2048": dict(B=16, H=64, N=2048, d=128, dtype="bf16", causal=False),
    "c3_fp8": dict(B=4, H=32, N=4096, d=128, dtype="fp8", causal=True),
    "c5_per_gpu": dict(B=16, H=8, N=16384, d=128, dtype="fp8", causal=False),   # BASELINE.json configs[4], one GPU's head shard
    "fp8_4k": dict(B=4, H=32, N=4096, d=128, dtype="fp8", causal=False),
    "fp8_1k": dict(B=16, H=32, N=1024, d=128, dtype="fp8", causal=False),
    "fp8_8k_causal": dict(B=8, H=8, N=8192, d=128, dtype="fp8", causal=True),
    "fp8_ragged": dict(B=8, H=8, N=8100, d=128, dtype="fp8", causal=False),
    "fp8_ragged_causal": dict(B=4, H=32, N=4000, d=128, dtype="fp8", causal=True),
    "fp8_2k_causal": dict(B=8, H=32, N=2048, d=128, dtype="fp8", causal=True),
}
TORCH_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "f32": torch.float32, "fp8": torch.float8_e4m3fn}
# Dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters" (TFLOP/s)
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "f32": 157.3, "fp8": 5000.0}  # fp8: the rate of v_mfma_f32_32x32x64_f8f6f4
# (default fp8 kernel mfma8x); the 32x32x16 fp8 form of variant mfma8 runs at the bf16 rate (2500)


def flops(c):
    f = 4.0 * c["B"] * c["H"] * c["N"] ** 2 * c["d"]
    return f * 0.5 if c["causal"] else f


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(c, budget_s=12.0):
    """The reference's CPU path -- torch SDPA(scale=1) on fp32 CPU tensors (src/test_correctness.py:33) -- on a bounded
    sample of the same workload (same N, d, causal; B=1 and a few heads), in both flavours the reference's bench times
    (src/bench.py:80,83-85): the default backend and the "naive" SDPBackend.MATH one."""
    ncores = os.cpu_count() or 1
    torch.set_num_threads(ncores)
    Hs = min(4, c["H"])
    sc = dict(c, B=1, H=Hs)
    g = torch.Generator().manual_seed(42)
    Q, K, V = (torch.randn(1, Hs, c["N"], c["d"], generator=g) for _ in range(3))
    fn = lambda: torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1.0, is_causal=c["causal"])

    def timed(f, budget):
        f()
        t0 = time.perf_counter()
        reps = 0
        while True:
            f()
            reps += 1
            el = time.perf_counter() - t0
            if el > budget or reps >= 50:
                return reps, el
    reps, el = timed(fn, budget_s * 0.5)
    gf = flops(sc) * reps / el / 1e9
    out = {"value": round(gf / 1e3, 5), "unit": "TFLOP/s", "cores": ncores, "cpu": cpu_model(), "kind": "reference",
           "sample": f"torch SDPA(scale=1) fp32 on CPU, default backend, B=1 H={Hs} N={c['N']} d={c['d']} "
                     f"causal={c['causal']}, {reps} reps in {el:.1f}s", "gflops": round(gf, 2)}
    try:  # the "naive" flavour: SDPBackend.MATH materialises the N x N scores (src/bench.py:83-85)
        from torch.nn.attention import SDPBackend, sdpa_kernel

        def fn_math():
            with sdpa_kernel(SDPBackend.MATH):
                return fn()
        reps_m, el_m = timed(fn_math, budget_s * 0.4)
        out["math_backend"] = {"value": round(flops(sc) * reps_m / el_m / 1e12, 5), "unit": "TFLOP/s",
                               "sample": f"same sample under SDPBackend.MATH, {reps_m} reps in {el_m:.1f}s"}
    except Exception as e:
        out["math_backend"] = {"error": str(e)[:100]}
    # configs[0] (c1) and configs[1] (c2) timed FULLY in both flavours, configs[2] (c3) at one full repetition
    # of the default backend (its MATH flavour materialises 4 x 32 x 4096^2 fp32 scores: the sample above stands for it)
    full = {}
    for name, cc, reps in (("c1", dict(B=1, H=2, N=128, d=64, causal=False), 200), ("c2", dict(B=2, H=8, N=1024, d=64, causal=False), 10),
                           ("c3", dict(B=4, H=32, N=4096, d=128, causal=True), 1)):
        try:
            g2 = torch.Generator().manual_seed(42)
            q, k, v = (torch.randn(cc["B"], cc["H"], cc["N"], cc["d"], generator=g2) for _ in range(3))
            f = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, scale=1.0, is_causal=cc["causal"])
            def timed_reps(fun, cap, budget=2.5):
                """mean ms over up to `cap` repetitions inside a time budget (256 host threads make even c1 cost ~0.1 s a call)"""
                fun() if cap > 1 else None          # one warm-up, except for the single full c3 repetition
                t0, n = time.perf_counter(), 0
                while n < cap and (n == 0 or time.perf_counter() - t0 < budget):
                    fun()
                    n += 1
                return (time.perf_counter() - t0) * 1e3 / n, n
            ms, n = timed_reps(f, reps)
            ent = {"shape": f"B={cc['B']} H={cc['H']} N={cc['N']} d={cc['d']} fp32 causal={cc['causal']}", "reps": n,
                   "ms": round(ms, 4), "gflops": round(flops(cc) / ms / 1e6, 2)}
            if name != "c3":
                from torch.nn.attention import SDPBackend, sdpa_kernel

                def f_math():
                    with sdpa_kernel(SDPBackend.MATH):
                        return f()
                ms_m, n_m = timed_reps(f_math, reps)
                ent["math_backend"] = {"ms": round(ms_m, 4), "gflops": round(flops(cc) / ms_m / 1e6, 2), "reps": n_m}
            full[name] = ent
            del q, k, v
        except Exception as e:
            full[name] = {"error": str(e)[:100]}
    out["full_configs"] = full
    # the oracle's scalar C port, one core, on a smaller slice (N^2 work: keep it to a few seconds)
    try:
        from oracle import fa2_oracle
        n = min(c["N"], 1024)
        q, k, v = (x[:, :1, :n].contiguous().numpy() for x in (Q, K, V))
        t1 = time.perf_counter()
        fa2_oracle.forward(q, k, v, "float32", causal=c["causal"], B_r=64, B_c=64)
        el1 = time.perf_counter() - t1
        f1 = 4.0 * n * n * c["d"] * (0.5 if c["causal"] else 1.0)
        out["port_1core_gflops"] = round(f1 / el1 / 1e9, 3)
        out["port_sample"] = f"oracle/fa2_oracle.c, 1 core, B=1 H=1 N={n} d={c['d']}"
    except Exception as e:  # the baseline leg must never take the bench down
        out["port_error"] = str(e)[:100]
    return out


def profiled_traffic(config, variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this very
    command (profiles/rNN/<config>_*_rocprof.json: FETCH_SIZE x 1024 x 2 -- the gfx950 half-count correction of
    MI355X_MICROARCH.md section HBM -- plus WRITE_SIZE x 1024; collected in separate --pmc passes by
    scripts/gpu_prof.sh).  bench.py cannot collect PMC counters itself; None if no summary is committed."""
    import glob
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{config}_*_rocprof.json")))
    files = [f for f in files if f"_{variant}_" in os.path.basename(f)] or files     # (the pass of the kernel that runs, if there is one)
    for f in files:
        try:
            with open(f) as fh:
                d = json.load(fh)
            best = (d["hbm_bytes_per_launch"]["total"], os.path.relpath(f, ROOT))
        except Exception:
            continue
    return best


def measured_peak(dtype):
    """MFMA-only rate of this dtype's matrix instruction measured on an MI355X with non-zero operands
    (scripts/probes/mfma_peak.hip -> profiles/rNN/roofline.json) -- context for `frac`, which stays against the vendor
    peak.  None if no probe result is committed."""
    import glob
    keys = {"bf16": ("bf16_random",), "fp16": ("f16_random", "bf16_random"), "fp8": ("fp8_random",), "f32": ("f32_random",)}.get(dtype, ())
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "roofline.json")), reverse=True):
        try:
            with open(f) as fh:
                d = json.load(fh)["mfma_only_tflops"]
            for key in keys:
                if key in d:
                    return {"tflops": d[key], "source": os.path.relpath(f, ROOT), "key": key}
        except Exception:
            continue
    return None


# BASELINE.json configs[3] / configs[4]: the whole problem, sharded over the heads (c4) or over batch x heads (c5)
STRONG = {2: dict(name="c4", B=8, H=32, N=8192, d=128, dtype="bf16", causal=False),
          4: dict(name="c4", B=8, H=32, N=8192, d=128, dtype="bf16", causal=False),
          8: dict(name="c5", B=16, H=64, N=16384, d=128, dtype="fp8", causal=False)}


def strong_scaling(c, world, rank, dev, sync_all, iters=5):
    """The same TOTAL problem on 1 GPU (rank 0 alone) and head-sharded over all ranks, with and without the all-gather of O."""
    import torch.distributed as dist
    from flash_attention_dlrs_amd import flash_attention_forward
    from flash_attention_dlrs_amd.sharded import flash_attention_forward_sharded
    dtype = TORCH_DTYPE[c["dtype"]]
    spread = 0.5 if c["dtype"] == "fp8" else 1.0
    Hs = c["H"] // world
    torch.manual_seed(1234 + rank)
    Q, K, V = ((torch.randn(c["B"], Hs, c["N"], c["d"], device=dev) * spread).to(dtype) for _ in range(3))

    def timed(fn):
        fn()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sync_all()
        t = torch.tensor([(time.perf_counter() - t0) / iters], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()
    t_shard = timed(lambda: flash_attention_forward(Q, K, V, dev, causal=c["causal"]))
    t_gather = timed(lambda: flash_attention_forward_sharded(Q, K, V, causal=c["causal"], gather=True))
    del Q, K, V
    t_one = torch.zeros(1, device=dev, dtype=torch.float64)
    if rank == 0:   # the whole problem on one GPU, the others idle
        Qf, Kf, Vf = ((torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev) * spread).to(dtype) for _ in range(3))
        flash_attention_forward(Qf, Kf, Vf, dev, causal=c["causal"])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(2):
            flash_attention_forward(Qf, Kf, Vf, dev, causal=c["causal"])
        torch.cuda.synchronize(dev)
        t_one[0] = (time.perf_counter() - t0) / 2
        del Qf, Kf, Vf
    dist.all_reduce(t_one, op=dist.ReduceOp.MAX)
    F = flops(dict(c))
    return {"config": f"{c['name']}: B={c['B']} H={c['H']} N={c['N']} d={c['d']} {c['dtype']}, {Hs} heads per GPU",
            "ms_1gpu_whole_problem": round(t_one.item() * 1e3, 3), "ms_sharded": round(t_shard * 1e3, 3),
            "ms_sharded_plus_gather": round(t_gather * 1e3, 3),
            "speedup_vs_1gpu": round(t_one.item() / t_shard, 3), "speedup_vs_1gpu_with_gather": round(t_one.item() / t_gather, 3),
            "tflops_aggregate": round(F / t_shard / 1e12, 1), "tflops_per_gpu": round(F / t_shard / 1e12 / world, 1),
            "tflops_aggregate_with_gather": round(F / t_gather / 1e12, 1)}


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: this process starts N fresh rank processes (one per GPU) through
    `python -m torch.distributed.run` as a CHILD, relays rank 0's JSON line and returns the children's worst exit code.  It runs
    before anything here touches the GPU: the parent never initialises HIP and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = 0
    for line in proc.stdout:
        try:
            is_result = "metric" in json.loads(line)
        except ValueError:
            is_result = False
        if is_result and lines == 0:
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write(f"bench.py: the rank processes printed {lines} result lines\n")
        return 1
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--gather", action="store_true", help="include the RCCL all-gather of O in the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the scaled-input and gather side measurements")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the rank processes from this process even for --gpus 1 (what --gpus N > 1 does when not under torch.distributed.run)")
    args = ap.parse_args()

    if "RANK" not in os.environ and (args.gpus > 1 or args.self_launch):
        sys.exit(self_launch(args.gpus))

    import torch.distributed as dist
    from flash_attention_dlrs_amd import _lib, flash_attention_forward
    from flash_attention_dlrs_amd.sharded import flash_attention_forward_sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but torch.distributed.run started {world} ranks")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ   # torch.distributed.run, also at world size 1
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    grouped = dist.is_initialized()

    c = CONFIGS[args.config]
    dtype = TORCH_DTYPE[c["dtype"]]
    torch.manual_seed(42 + rank)  # src/bench.py:26; rank r draws its own head shard
    Q, K, V = (torch.randn(c["B"], c["H"], c["N"], c["d"], device=dev).to(dtype) for _ in range(3))
    variant = args.variant

    def step():
        if args.gather and grouped:
            return flash_attention_forward_sharded(Q, K, V, causal=c["causal"], gather=True)
        return flash_attention_forward(Q, K, V, dev, causal=c["causal"], variant=variant)

    def sync_all():
        torch.cuda.synchronize(dev)
        if grouped:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # Clock ramp: an idle MI355X runs its first milliseconds of work below its sustained clock (measured:
    # 751 TFLOP/s with 5 warm-up steps vs 831 with 50 on the same device).  Spin ~0.25 s of the same kernel
    # before the W untimed warm-up steps so that short (K, W) choices measure the steady state too.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < float(os.environ.get("FA2_BENCH_SPIN", "0.25")):
        for _ in range(20):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    # The timed region: exactly K steps between two HIP events on the stream the kernel is launched on (torch's current
    # stream).  One event PAIR PER STEP, as in round 1, put two marker packets between consecutive launches -- 9.5 us of
    # idle GPU per 457 us kernel, i.e. the measurement cost 2 % of `value`; the average launch duration is the bracket
    # divided by K (it includes the launch-to-launch gap, so it can only overstate the kernel).  Per-launch events for
    # the median / minimum run in a second, untimed pass of the same K steps.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sync_all()
    # (a blocking synchronize hands control back tens of microseconds after the GPU has gone idle, and an idle GPU takes as long
    # again to pick up the next packet: one more launch, a spin on its event, and the timed steps follow within microseconds)
    step()
    evw = torch.cuda.Event()
    evw.record()
    while not evw.query():
        pass
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    while not ev1.query():      # spin on the closing event: a blocking synchronize wakes the host tens of microseconds after the last
        pass                    # kernel has ended -- 0.3-0.5 % of a 20-step bracket of 0.44-ms kernels; the synchronize below then returns at once
    sync_all()
    el = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    kern_avg = ev0.elapsed_time(ev1) / args.steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for s, e in ev:
        s.record()
        step()
        e.record()
    torch.cuda.synchronize(dev)
    kern_ms = sorted(s.elapsed_time(e) for s, e in ev)

    F = flops(c)
    ms_per_step = el * 1e3 / args.steps
    value = world * F / (el / args.steps) / 1e12
    peak = PEAK_TFLOPS[c["dtype"]]
    achieved = F / (kern_avg * 1e-3) / 1e12
    tile = _lib.query_tile(c["N"], c["d"], {"bf16": 2, "fp16": 1, "f32": 0, "fp8": 4}[c["dtype"]], c["causal"], B=c["B"], H=c["H"])

    extras = {}
    if not args.no_extras:
        # (1) inputs scaled by d^-1/4 (== the usual 1/sqrt(d)): softmax no longer nearly one-hot
        s4 = c["d"] ** -0.25
        Q2, K2 = (Q.float() * s4).to(dtype), (K.float() * s4).to(dtype)
        for _ in range(30):
            flash_attention_forward(Q2, K2, V, dev, causal=c["causal"], variant=variant)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        a.record()
        for _ in range(100):
            flash_attention_forward(Q2, K2, V, dev, causal=c["causal"], variant=variant)
        b.record()
        torch.cuda.synchronize(dev)
        extras["tflops_inputs_scaled_d^-1/4"] = round(F / (a.elapsed_time(b) / 100 * 1e-3) / 1e12, 2)
        del Q2, K2
        # (1a) the same shape and inputs WITHOUT the causal mask (BASELINE.json's metric string and the north star's target
        # name the shape, configs[2] adds the mask: `value` is the masked case, this is the other reading)
        if c["causal"]:
            for _ in range(10):
                flash_attention_forward(Q, K, V, dev, causal=False, variant=variant)
            torch.cuda.synchronize(dev)
            a.record()
            for _ in range(50):
                flash_attention_forward(Q, K, V, dev, causal=False, variant=variant)
            b.record()
            torch.cuda.synchronize(dev)
            tnc = 2.0 * F / (a.elapsed_time(b) / 50 * 1e-3) / 1e12
            extras["same_shape_no_mask"] = {"tflops": round(tnc, 2), "pct_of_mfma_peak": round(100 * tnc / peak, 2)}
        # (1b) the backward of the same workload (the reference bench's default mode,
        # src/bench.py:20): D + dQ + dK/dV launches of include/fa2_bwd.h, TFLOP/s at the 2.5 x forward convention
        if c["dtype"] in ("bf16", "fp16", "f32"):
            from flash_attention_dlrs_amd import flash_attention_backward
            O, Ls = flash_attention_forward(Q, K, V, dev, causal=c["causal"])
            dO = torch.randn_like(Q)
            for _ in range(5):
                flash_attention_backward(Q, K, V, O, dO, Ls, dev, causal=c["causal"])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            a.record()
            for _ in range(20):
                flash_attention_backward(Q, K, V, O, dO, Ls, dev, causal=c["causal"])
            b.record()
            torch.cuda.synchronize(dev)
            tb = a.elapsed_time(b) / 20
            extras["backward"] = {"ms": round(tb, 4), "tflops_2.5x_fwd_convention": round(2.5 * F / (tb * 1e-3) / 1e12, 2),
                                  "launches": "D, dQ (query-block owner), dK/dV (key-block owner); deterministic"}
            del O, Ls, dO
        # (2) the optional exchange step: all-gather of the O shards over xGMI, overlapped per batch element
        if grouped:
            for _ in range(2):
                flash_attention_forward_sharded(Q, K, V, causal=c["causal"], gather=True)
            sync_all()
            t1 = time.perf_counter()
            for _ in range(5):
                flash_attention_forward_sharded(Q, K, V, causal=c["causal"], gather=True)
            sync_all()
            tg = (time.perf_counter() - t1) / 5
            o_bytes = Q.numel() * Q.element_size()
            extras["gather"] = {"included_in_step": bool(args.gather), "ms_compute_plus_gather": round(tg * 1e3, 4),
                                "o_shard_MiB": o_bytes / 2 ** 20, "world_size": world,
                                "tflops_with_gather": round(world * F / tg / 1e12, 2)}
        # (3) the strong-scaling configurations BASELINE.json names for this GPU count: configs[3] on 4 GPUs, configs[4] on 8
        if world in STRONG and grouped:
            extras["strong_scaling"] = strong_scaling(STRONG[world], world, rank, dev, sync_all)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(c)

    picked = {v: k for k, v in _lib.VARIANTS.items()}.get(tile[0], variant)      # the name of the kernel the table picked
    traffic = profiled_traffic(args.config, picked) if args.variant == "auto" else None
    if rank == 0:
        out = {
            "metric": f"attention fwd TFLOP/s per GPU (B={c['B']},H={c['H']},N={c['N']},d={c['d']} {c['dtype']}); % MFMA peak",
            "value": round(value, 3), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": c["dtype"], "data": "synthetic",
            "config": {"workload": f"{args.config}: FA-2 forward B={c['B']} H={c['H']} N={c['N']} d={c['d']} "
                                   f"{c['dtype']} causal={c['causal']} scale=1 per GPU; seed 42+rank N(0,1)",
                       "kernel_variant": variant, "tile": {"variant": tile[0], "B_r": tile[1], "B_c": tile[2],
                                                           "waves": tile[3]},
                       "sharding": "heads" if world > 1 else "none",
                       "flops_per_step_per_gpu": F, "pct_of_mfma_peak": round(100 * value / world / peak, 2)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic[0] if traffic else None,
                         "traffic_source": traffic[1] if traffic else None,
                         "algorithmic_bytes": (4 * c["d"] + 1) * c["B"] * c["H"] * c["N"] * Q.element_size(),
                         "kernel_ms_avg": round(kern_avg, 5), "kernel_ms_median": round(kern_ms[len(kern_ms) // 2], 5),
                         "kernel_ms_min": round(kern_ms[0], 5),
                         "measured_mfma_only_peak": measured_peak(c["dtype"])},
            "cpu_baseline": cpu,
            "extras": extras,
            "n_ranks_seen_by_rccl": dist.get_world_size() if grouped else None,
            "lib": _lib.version(),
        }
        print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
