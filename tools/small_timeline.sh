#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of ONE lone MSM per length: where a short sum's time goes.
#   tools/small_timeline.sh <out-dir> <log2n>...      (repository root, GPU box)
set -e
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for l in "$@"; do
  d=$(mktemp -d /tmp/prof.XXXX)
  rocprofv3 --kernel-trace -d "$d" -o run -- python3 bench.py --log2n $l --in-flight 1 --no-extras --no-cpu-baseline --steps 4 --warmup 2 > "$out/tl_$l.json" 2> "$out/tl_$l.err" || { tail -5 "$out/tl_$l.err"; exit 1; }
  db=$(find "$d" -name '*.db' | head -1)
  python3 tools/rocpd_timeline.py "$db" --anchor k_sort_count --nth -1 --before 0.05 --after 3 --min-ms 0 > "$out/tl_$l.txt"
  rm -rf "$d"
done
