"""The shipped gfx950 code object, read without a GPU: the hot kernels spill nothing and hold their field
arithmetic inline.  (The lane-pair mixed addition once became an out-of-line call when a second kernel started to
share it -- every test stayed green and the G2 sum went from 8.2 to 11.2 ms.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "playsnark_amd", "libplaysnark_hip.so")


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    if not (os.path.exists(LIB) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip("library or LLVM tools not present")
    d = tmp_path_factory.mktemp("co")
    shutil.copy(LIB, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
    co = [f for f in os.listdir(d) if f.endswith("gfx950")]
    assert len(co) == 1, os.listdir(d)
    return str(d / co[0])


def kernel_notes(co):
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, {}
    for line in out.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "agpr_count" and cur.get("name"):
            kernels[cur["name"]] = cur
            cur = {}
        cur[key] = val
    if cur.get("name"):
        kernels[cur["name"]] = cur
    return kernels


def test_hot_kernels_do_not_spill(code_object):
    notes = kernel_notes(code_object)
    hot = [n for n in notes if re.search(r"k_accumulate|k_reduce_sum|k_ntt_passILb0|k_table_next|k_fixed_base_mul|k_batch_to_affine", n)]
    assert len(hot) >= 9, hot
    for n in hot:
        # the lane-pair G2 accumulation runs two waves per SIMD on purpose (msm.hpp, PS_G2_ACC_WAVES): 46 spilled registers
        limit = 48 if ("k_accumulate" in n and "Fp2s" in n) else 0
        assert int(notes[n]["vgpr_spill_count"]) <= limit, (n, notes[n])


def test_tail_kernels_fit_beside_an_accumulation_wave(code_object):
    """The chains of the long sums' tail (fix-up, 8-bucket running sums, pyramid) are capped at 256 registers
    (msm.hpp, PS_TAIL_WAVES = 2) and pay for it in spills: with sums in flight their waves must fit the half of a SIMD's
    register file that a retiring accumulation wave frees (k_reduce_l1 at 306 registers waited for a whole SIMD to drain:
    G2 sum in flight 8.0 -> 7.3-7.6 ms, PHGR13 at 2^20 29.5 -> 28.7 ms)."""
    notes = kernel_notes(code_object)
    tail = [n for n in notes if re.search(r"k_reduce_l1|k_fixupI|k_reduce_pyr", n)]
    assert len(tail) == 6, tail
    for n in tail:
        assert int(notes[n]["vgpr_count"]) + int(notes[n]["agpr_count"]) <= 256, (n, notes[n])


def test_key_conversion_kernels_of_g1_run_two_waves_per_simd(code_object):
    """k_ec_ntt_stage / k_ec_scale (csrc/lagrange.hpp: a GLV scalar multiplication per thread, its point operations inlined)
    are latency chains of dependent multiply-adds: the G1 instances must stay within 256 registers without spilling so that
    two waves share a SIMD (2^20-constraint key: 11.2 -> 8.6 s per G1 array)."""
    notes = kernel_notes(code_object)
    ec = [n for n in notes if re.search(r"k_ec_ntt_stageINS_2FpE|k_ec_scaleINS_2FpE", n)]
    assert len(ec) == 3, ec
    for n in ec:
        assert int(notes[n]["vgpr_count"]) + int(notes[n]["agpr_count"]) <= 256, (n, notes[n])
        assert int(notes[n]["vgpr_spill_count"]) == 0, (n, notes[n])


def test_accumulation_kernels_hold_the_mixed_addition_inline(code_object):
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code_object], check=True, capture_output=True, text=True).stdout
    bodies = {}
    name = None
    for line in asm.splitlines():
        m = re.match(r"[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            bodies[name] = 0
        elif name and "v_mad_" in line and "64" in line:
            bodies[name] += 1
    acc = {n: c for n, c in bodies.items() if "k_accumulate" in n}
    assert len(acc) == 2, acc
    for n, c in acc.items():
        # one mixed addition is 3542 multiply-adds per lane for G1 and 5488 for the lane-pair G2 kernel;
        # the prefetching loop keeps one copy
        assert c >= 3500, (n, c)
