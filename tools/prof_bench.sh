#!/bin/bash
# rocprofv3 kernel trace of one bench.py configuration -> kernel-stats CSV (tools/rocpd_stats.py).
#   tools/prof_bench.sh <out-prefix> <bench.py args...>
# Run from the repository root on the GPU box; writes <out-prefix>.csv and <out-prefix>.json (the bench line).
set -e
out=$1; shift
d=$(mktemp -d /tmp/prof.XXXX)
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$d" -o run -- python3 bench.py "$@" > "$out.json" 2> "$out.err" || { tail -5 "$out.err"; exit 1; }
db=$(find "$d" -name '*.db' | head -1)
python3 tools/rocpd_stats.py "$db" > "$out.csv"
rm -rf "$d"
