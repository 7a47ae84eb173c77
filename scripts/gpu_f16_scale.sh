#!/bin/bash
# f16 with the usual softmax scale 1 / sqrt(d) (scores of sigma ~1.4 log2 units: the running maximum rarely moves): a64 against a16
set -u
cd "$(dirname "$0")/.."
timeout -k 10 400 python benchmarks/variants.py --rounds 7 --iters 10 --pairs ref_bench:a64:SCALE=0.0884,ref_bench:a16:SCALE=0.0884,ref_bench:a64:SCALE=0.25,ref_bench:a16:SCALE=0.25,ref_bench:a64:SCALE=0.5,ref_bench:a16:SCALE=0.5,ref_bench:a64,ref_bench:a16 2>&1 | grep pair
