#!/bin/bash
# Host-only entry points of the product (pairing, subgroup checks, ps_points_sum, ps_point_convert, ps_points_lincomb)
# against the ASan + UBSan build of the library.  CPU only.  Usage: tools/run_host_sanitizers.sh [log file]
set -e
cd "$(dirname "$0")/.."
make -C playsnark_amd/csrc sanitized
export LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libplaysnark_hip_san.so
python -m pytest tests/test_verify_pairing.py tests/test_abi.py -q -m "not gpu" -k "host_only or subgroup or host_fold or point_convert" 2>&1 | tee "${1:-/dev/stdout}"
