#!/usr/bin/env python3
"""Counterpart of the reference's src/bench.py (modes "fwd" and "bwd", src/bench.py:20,93-97): sweep N = 2^7 .. 2^15 at B=8, H=16,
d=128, fp16 (src/bench.py:8-18), time every provider with do_bench semantics (warm-up ~25 ms, ~100 ms of
timed repetitions, HIP events per repetition, a cache flush between repetitions; src/bench.py:61-62,99),
and write the CSV the reference's plotting script reads:

    bench_out/fused-attention-B{B}-H{H}-d{d}-fwd-{dtype}.csv      (src/bench.py:47, src/plot_bench_results.py:41-57)

with an `N` column and one column of mean milliseconds per provider display name.  Providers: this
repository's kernel, and the reference's torch providers under the reference's own display names
(src/bench.py:38-41,76-85): "Torch FA-2" (SDPBackend.FLASH_ATTENTION), "Torch xFormers"
(SDPBackend.EFFICIENT_ATTENTION), "Torch Math" (SDPBackend.MATH), plus torch's default dispatch.  A backend
this torch build cannot run for the shape gives NaN, as an out-of-memory does in the reference
(src/bench.py:100-110).  The competitor providers of the reference that are CUDA wheels (flash-attn, the
vendored OpenAI tutorial) do not exist on ROCm and are omitted.
A second file `...-tflops.csv` carries the same sweep as TFLOP/s (4*B*H*N^2*d / t; backward: 2.5 x that,
the convention of the vendored tutorial, src/flash_attention_openai_tutorial.py:630-635).  Mode "bwd" times
`O.backward(dO, retain_graph=True)` exactly as the reference does (src/bench.py:93-96).
"""
import argparse
import csv
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import FlashAttention  # noqa: E402

B, H, d = 8, 16, 128          # src/bench.py:8-10
N_MIN_log, N_MAX_log = 7, 15  # src/bench.py:11-12
BENCH_DIR = "bench_out"       # src/bench.py:14
DTYPE = torch.float16         # src/bench.py:18


def do_bench(fn, warmup_ms=25, rep_ms=100):
    """triton.testing.do_bench semantics: estimate the run time, then n_warmup / n_repeat from the budgets,
    flush the caches (a 256 MiB write) before every timed repetition, return the mean in ms."""
    dev = torch.device("cuda")
    cache = torch.empty(256 << 20, dtype=torch.int8, device=dev)
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        cache.zero_()
        fn()
    b.record()
    torch.cuda.synchronize()
    est = a.elapsed_time(b) / 5
    n_warm = max(1, int(warmup_ms / est))
    n_rep = max(1, int(rep_ms / est))
    for _ in range(n_warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_rep)]
    for s, e in ev:
        cache.zero_()
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return sum(s.elapsed_time(e) for s, e in ev) / n_rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-max-log", type=int, default=N_MAX_log)
    ap.add_argument("--providers", default="hip,torch-sdpa,torch-fa,torch-xformers,torch-math")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "bwd"])
    ap.add_argument("--out-dir", default=BENCH_DIR)
    args = ap.parse_args()
    torch.manual_seed(42)  # src/bench.py:26
    gpu = torch.device("cuda")
    dtype_str = str(DTYPE).split(".")[1]
    names = {"hip": f"MI355X HIP FA-2 [{dtype_str.upper()}]", "torch-sdpa": f"Torch SDPA default [{dtype_str.upper()}]",
             "torch-fa": f"Torch FA-2 [{dtype_str.upper()}]", "torch-xformers": f"Torch xFormers [{dtype_str.upper()}]",
             "torch-math": f"Torch Math [{dtype_str.upper()}]"}
    backends = {"torch-fa": "FLASH_ATTENTION", "torch-xformers": "EFFICIENT_ATTENTION", "torch-math": "MATH"}
    providers = args.providers.split(",")
    rows = []
    for N in [2 ** i for i in range(N_MIN_log, args.n_max_log + 1)]:
        row = {"N": float(N)}
        try:
            Q, K, V = (torch.randn(B, H, N, d, dtype=DTYPE, device=gpu, requires_grad=args.mode == "bwd") for _ in range(3))
        except torch.cuda.OutOfMemoryError:
            break
        for p in providers:
            if p == "hip":
                fn = lambda: FlashAttention.apply(Q, K, V)
            elif p == "torch-sdpa":
                fn = lambda: torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1)
            else:
                def fn(backend=getattr(torch.nn.attention.SDPBackend, backends[p])):
                    with torch.nn.attention.sdpa_kernel(backend):
                        return torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1)
            try:
                if p == "torch-math" and N > (4096 if args.mode == "bwd" else 8192):
                    raise torch.cuda.OutOfMemoryError()  # N^2 score matrix: 2^15 needs 256 GiB in fp16
                if args.mode == "bwd":  # src/bench.py:93-96
                    O = fn()
                    dO = torch.randn_like(O)
                    bench_fn = lambda: O.backward(dO, retain_graph=True)
                else:
                    bench_fn = fn
                ms = do_bench(bench_fn)
            except (torch.cuda.OutOfMemoryError, RuntimeError) as e:  # reference: NaN on OOM (src/bench.py:100-110)
                ms = float("nan")
            row[names[p]] = ms
            print(f"Benchmarking {args.mode} (N={N}, H={H}, B={B}, d={d}) for {p} ... {ms:.4f} ms", flush=True)
        rows.append(row)
    os.makedirs(args.out_dir, exist_ok=True)
    base = os.path.join(args.out_dir, f"fused-attention-B{B}-H{H}-d{d}-{args.mode}-{dtype_str}")
    cols = ["N"] + [names[p] for p in providers]
    with open(base + ".csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        w.writerows(rows)
    with open(base + "-tflops.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        for r in rows:
            fl = (10.0 if args.mode == "bwd" else 4.0) * B * H * r["N"] ** 2 * d
            w.writerow({c: (r[c] if c == "N" else fl / (r[c] * 1e-3) / 1e12) for c in cols})
    print(open(base + "-tflops.csv").read())


if __name__ == "__main__":
    main()
