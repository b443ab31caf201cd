#!/bin/bash
# The CPU test suite (-m "not gpu") against the AddressSanitizer + UBSan builds of the product's host code and of the oracle.
#   make -C raytracing-in-a-weekend_amd/csrc host-asan && make -C oracle asan && scripts/run_cpu_tests_asan.sh
set -e
cd "$(dirname "$0")/.."
make -s -C raytracing-in-a-weekend_amd/csrc host-asan
make -s -C oracle asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export RTW_HIP_LIB=$PWD/raytracing-in-a-weekend_amd/librtw_hip_asan.so
export RTW_ORACLE_LIB=$PWD/oracle/librtw_oracle_asan.so
python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
