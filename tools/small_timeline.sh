#!/bin/bash
# Kernel timeline (rocprofv3 --kernel-trace) of ONE lone MSM per length: where a short sum's time goes.
#   [GROUP=g2] tools/small_timeline.sh <out-dir> <log2n>...      (repository root, GPU box)
set -e
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for l in "$@"; do
  d=$(mktemp -d /tmp/prof.XXXX)
  rocprofv3 --kernel-trace -d "$d" -o run -- python3 bench.py --group ${GROUP:-g1} --log2n $l --in-flight 1 --no-extras --no-cpu-baseline --steps 4 --warmup 2 > "$out/tl_${GROUP:-g1}_$l.json" 2> "$out/tl_${GROUP:-g1}_$l.err" || { tail -5 "$out/tl_${GROUP:-g1}_$l.err"; exit 1; }
  db=$(find "$d" -name '*.db' | head -1)
  python3 tools/rocpd_timeline.py "$db" --anchor k_sort_count --nth -1 --before 0.05 --after 5 --min-ms 0 > "$out/tl_${GROUP:-g1}_$l.txt"
  rm -rf "$d"
done
