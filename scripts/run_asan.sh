#!/bin/bash
# CPU sanitizer run: the host side of libfa2_hip.so and the oracle, built with AddressSanitizer + UBSan, under
# the CPU tests that exercise them -- argument validation and the tile table through the C ABI (tests/test_abi.py), the oracle
# against the golden vectors (tests/test_oracle.py).  GPU sanitizers are not available on the pool: this is the CPU build only.
set -eu
cd "$(dirname "$0")/.."
make -C flash_attention_dlrs_amd/csrc asan -j8 > /tmp/fa2_asan_build.log 2>&1 || { tail -20 /tmp/fa2_asan_build.log; exit 2; }
make -C oracle asan > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1
export FA2_HIP_LIB=$PWD/flash_attention_dlrs_amd/libfa2_hip_asan.so FA2_ORACLE_LIB=$PWD/oracle/libfa2_oracle_asan.so
python -m pytest tests/test_abi.py tests/test_oracle.py tests/test_oracle_bwd.py -q -m "not gpu" -p no:cacheprovider "$@"
