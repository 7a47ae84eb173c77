#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/gpu_prof.sh (gpurun_out/prof/{stats,pmc_*}) into the tracked summary
profiles/rNN/<tag>_rocprof.json + <tag>_kernel_stats.csv.

    python scripts/summarize_prof.py gpurun_out/prof profiles/r01 c3_mfma16d mfma16d_kernel "c3: ..."

HBM bytes: FETCH_SIZE (KiB) x 1024 x 2 (gfx950 counts 128-B requests at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE x 1024.
"""
import csv
import glob
import json
import os
import shutil
import sys


def newest_run(files):
    """gpurun merges new output into the local gpurun_out/: keep only the files of the most recent profiler process per directory
    (rocprofv3 prefixes its files with its process id)"""
    files = list(files)
    if not files:
        return files
    pid = os.path.basename(max(files, key=os.path.getmtime)).split("_")[0]
    return [f for f in files if os.path.basename(f).split("_")[0] == pid]


def main():
    src, dst, tag, kfilter, workload = sys.argv[1:6]
    os.makedirs(dst, exist_ok=True)
    counters, kern_ns = {}, {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        vals, durs = {}, []
        for f in newest_run(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if kfilter in r["Kernel_Name"]:
                    vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in newest_run(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if kfilter in r["Kernel_Name"]:
                    durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in vals.items():
            counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
        if durs:
            kern_ns[os.path.basename(d)] = sum(durs) / len(durs)
    stats = None
    for f in newest_run(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(f)):
            if kfilter in r["Name"]:
                stats = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                         "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"])}
                break
    out = {"command": "rocprofv3 --kernel-trace [--stats | --pmc <group>] --output-format csv -- python3 bench.py " +
                      os.environ.get("BENCH_ARGS", "--steps 20 --warmup 5 --no-cpu-baseline --no-extras") + "  (scripts/gpu_prof.sh)",
           "workload": workload, "counters": counters, "kernel_ns_in_pmc_runs": kern_ns, "kernel_stats": stats}
    c = lambda k: counters.get(k, {}).get("mean_per_launch")
    if c("FETCH_SIZE") is not None and c("WRITE_SIZE") is not None:
        rd, wr = c("FETCH_SIZE") * 1024 * 2, c("WRITE_SIZE") * 1024
        out["hbm_bytes_per_launch"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
    der = {}
    if c("GRBM_GUI_ACTIVE") and stats:
        dur = kern_ns.get(next((k for k in kern_ns if "SQ_WAVE" in k), ""), stats["avg_ns"])
        der["clock_GHz"] = c("GRBM_GUI_ACTIVE") / 8 / dur
        if c("SQ_VALU_MFMA_BUSY_CYCLES"):
            der["mfma_busy_frac"] = c("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * der["clock_GHz"] * dur)
    if c("SQ_WAVE_CYCLES"):
        der["wave_cycle_split"] = {k: c(k) / c("SQ_WAVE_CYCLES") for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY") if c(k)}
    if c("TCC_HIT_sum") and c("TCC_MISS_sum"):
        der["l2_hit_rate"] = c("TCC_HIT_sum") / (c("TCC_HIT_sum") + c("TCC_MISS_sum"))
    if c("SQ_LDS_BANK_CONFLICT") is not None:
        der["lds_bank_conflict_cycles"] = c("SQ_LDS_BANK_CONFLICT")
    out["derived"] = der
    with open(os.path.join(dst, f"{tag}_rocprof.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({"stats": stats, "derived": der, "hbm": out.get("hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
