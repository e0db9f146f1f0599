#!/bin/bash
# sums in flight: queue depth A/B on one box (library built with -DPS_MSM_QUEUE=4)
export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libps_tailtune.so
for round in 1 2; do for l in 10 13 16 18 20; do for d in 2 3 4; do
echo "2^$l depth $d | $(python3 tools/small_sums_inflight.py $l $d 2>&1 | tail -n 1)"
done; done; done
