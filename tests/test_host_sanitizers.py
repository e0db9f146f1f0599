"""CPU: the lazy-limb contract of the product's field / curve / pairing arithmetic, machine-checked.  The same sources the
GPU kernels are built from (playsnark_amd/csrc/field.hpp, curve.hpp, pairing_math.inc) are compiled for the host with
AddressSanitizer + UndefinedBehaviorSanitizer (signed 64-bit column sums, shifts, array bounds) and driven with worst-case
limb-class operands, random lazy operands, bucket-style addition chains, the NTT butterfly sequences and the Miller loop
(tests/host_limb_check.cpp).  GPU sanitizers are not available on the pool; sanitizers run on the CPU build only.

The whole library under ASan / UBSan (host side of capi.hip: ps_points_sum, ps_point_convert, the verifiers' pairing code)
is a nine-minute hipcc build: `make -C playsnark_amd/csrc sanitized`, then tools/run_host_sanitizers.sh; its log is
committed under profiles/.  When that library is present the second test below runs the host-only entry points against it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_limb_contract_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "host_limb_check")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
                            os.path.join(ROOT, "tests", "host_limb_check.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    for line in ("fp products", "NTT butterfly sequences ok", "G1 group law ok", "G2 group law ok", "host field == device field", "device representation: pairing bilinearity ok, fast pairing path == literal path", "host field: pairing bilinearity ok, fast pairing path == literal path", "host_limb_check ok"):
        assert line in run.stdout


def test_host_entry_points_against_the_sanitized_library():
    lib = os.path.join(ROOT, "playsnark_amd", "libplaysnark_hip_san.so")
    if not os.path.exists(lib):
        pytest.skip("make -C playsnark_amd/csrc sanitized (a nine-minute build) has not been run")
    asan = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", PLAYSNARK_HIP_LIB=lib)
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "not gpu", "-k", "host_only or subgroup or host_fold or point_convert",
                          os.path.join(ROOT, "tests", "test_verify_pairing.py"), os.path.join(ROOT, "tests", "test_abi.py")],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
