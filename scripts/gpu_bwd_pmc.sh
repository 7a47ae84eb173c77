#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bwd
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 $R/benchmarks/bench_bwd.py --configs ${CFG:-c3_noncausal} > $OUT/pmc1.log 2>&1 || { tail -5 $OUT/pmc1.log; exit 2; }
f=$(find $OUT/pmc1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "bwd_" not in k: continue
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k)
    for c, v in sorted(m.items()):
        print(f"   {c:28s} {v:16.0f}  {v / wc:6.3f} of wave cycles")
PY
