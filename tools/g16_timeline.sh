#!/bin/bash
# Kernel timeline (all streams) of the last proof of tools/${WHAT:-g16}_experiment.py (WHAT=g16 or phgr13) at 2^LOG2N constraints.
#   tools/g16_timeline.sh <out-file> <log2n> [min-ms]
set -e
export TMPDIR=/tmp
d=$(mktemp -d /tmp/prof.XXXX)
LOG2N=$2 REPS=3 CIRCUIT=${CIRCUIT:-} rocprofv3 --kernel-trace -d "$d" -o run -- python3 tools/${WHAT:-g16}_experiment.py > "$1.log" 2>/dev/null
db=$(find "$d" -name '*.db' | head -1)
python3 tools/rocpd_timeline.py "$db" --anchor k_fr_to_mont --nth -2 --before 0.05 --after ${AFTER:-4} --min-ms ${3:-0} > "$1"
rm -rf "$d"
