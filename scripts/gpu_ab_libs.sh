#!/bin/bash
# A/B two builds of the library (separate processes, interleaved 3 times)
for r in 1 2 3; do
  for lib in libfa2_hip_base.so libfa2_hip.so; do
    echo "== $lib"
    FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/$lib python benchmarks/variants.py --pairs ${PAIRS:-c3_noncausal:auto,c3:auto} --rounds 5 2>&1 | grep pair
  done
done
