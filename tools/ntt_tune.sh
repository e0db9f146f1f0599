#!/bin/bash
# Pass shapes of the register-resident NTT (csrc/ntt.hpp, -DPS_NTT_PASS8: measured, not shipped) against each other.
# Measurement builds: hipcc ... -DPS_NTT_PASS8 -DPS_NTT_TUNE -o playsnark_amd/libps_tune.so capi.hip; libps_nobf.so adds
# -DPS_NTT8_SKIP_BF (no butterflies: what the memory side of a pass costs alone).  Per setting: the quotient's time inside Groth16Prove
# with the reference's key form at 2^20 constraints, and the passes by launch shape.
#   tools/ntt_tune.sh <out-dir> <lib-name> "<MAXK> <TILE>" ...
set -e
out=$1; lib=$2; shift 2
mkdir -p "$out"
export TMPDIR=/tmp
export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/$lib
for cfg in "$@"; do
  set -- $cfg
  export PS_NTT_MAXK=$1 PS_NTT_TILE=$2
  d=$(mktemp -d /tmp/prof.XXXX)
  MONOMIAL=1 REPS=2 rocprofv3 --kernel-trace -d "$d" -o run -- python3 tools/g16_experiment.py > "$out/tune_${lib}_$1_$2.log" 2>&1 || { tail -3 "$out/tune_${lib}_$1_$2.log"; continue; }
  python3 tools/ntt_pass_times.py "$(find "$d" -name '*.db' | head -1)" > "$out/tune_${lib}_$1_$2.txt"
  echo "== $lib maxk=$1 tile=$2: $(grep 'groth16 ms' "$out/tune_${lib}_$1_$2.log")"
  head -5 "$out/tune_${lib}_$1_$2.txt" | tail -4
  rm -rf "$d"
done
