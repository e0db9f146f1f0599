#!/bin/bash
# Kernel timeline of sums IN FLIGHT (three pending, all streams): what overlaps what.
#   tools/inflight_timeline.sh <out-file> <log2n> [window-ms]
set -e
export TMPDIR=/tmp
d=$(mktemp -d /tmp/prof.XXXX)
rocprofv3 --kernel-trace -d "$d" -o run -- python3 tools/small_sums_inflight.py $2 > "$1.log" 2>&1
db=$(find "$d" -name '*.db' | head -1)
python3 tools/rocpd_timeline.py "$db" --anchor k_sort_count --nth -12 --before 0.02 --after ${3:-1.6} --min-ms 0 > "$1"
rm -rf "$d"
