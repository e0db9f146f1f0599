#!/bin/bash
# Short sums after a change: the tail-mode parity tests, then one lone-sum kernel timeline per length and the in-flight /
# alone figures without event timing (extras-style).   tools/small_check.sh <out-dir> [log2n...]
set -e
out=$1; shift
lens=${@:-10 13 16}
mkdir -p "$out"
timeout -k 10 600 python3 -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "tail_of_a_sum or small_vs_reference or edge_scalars or window_size or skewed or config2 or g2_msm_2pow12" > "$out/pytest.log" 2>&1 || { tail -20 "$out/pytest.log"; exit 1; }
tail -2 "$out/pytest.log"
tools/small_timeline.sh "$out" $lens
python3 tools/small_sums.py $lens > "$out/small_sums.txt"
cat "$out/small_sums.txt"
