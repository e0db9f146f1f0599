#!/bin/bash
# Groth16Prove (Lagrange-form key): the split form of C (B in G1 as a sum of its own, s A + r B1 on the host) against the
# single sum with scalars s a_j + r b_j, by circuit size, on ONE box.   tools/g16_split_ab.sh [log2n ...]
for l in "$@"; do
  for mode in split single; do
    if [ $mode = split ]; then export PS_G16_B1_MIN_N=2; else export PS_G16_B1_MIN_N=1000000000; fi
    echo "2^$l $mode toy  | $(LOG2N=$l REPS=6 python3 tools/g16_experiment.py 2>&1 | grep 'groth16 ms')"
    echo "2^$l $mode bits | $(CIRCUIT=bits LOG2N=$l REPS=6 python3 tools/g16_experiment.py 2>&1 | grep 'groth16 ms')"
  done
done
