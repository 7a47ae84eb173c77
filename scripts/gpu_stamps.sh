#!/bin/bash
set -u
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_stamps.so
timeout -k 10 200 python benchmarks/stamps.py c3_noncausal 2>&1 | tail -9 | tee gpurun_out/stamps_nc.log
timeout -k 10 200 python benchmarks/stamps.py c3 2>&1 | tail -9 | tee gpurun_out/stamps_c.log
