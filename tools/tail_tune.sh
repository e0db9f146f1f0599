#!/bin/bash
# In-flight A/B of the short sums' tail on ONE box (measurement build: -DPS_TAIL_TUNE reads PS_T_* from the environment).
#   tools/tail_tune.sh <log2n> "LPB:NP:MSHIFT" ...      ("-" leaves a knob at its default)
export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libps_tailtune.so
l=$1; shift
for round in 1 2; do
  for ent in "$@"; do
    IFS=: read -r lpb np ms <<< "$ent"
    unset PS_T_LPB_BUSY PS_T_NP_BUSY PS_T_MSHIFT
    [ "$lpb" != "-" ] && export PS_T_LPB_BUSY=$lpb
    [ "$np" != "-" ] && export PS_T_NP_BUSY=$np
    [ "$ms" != "-" ] && export PS_T_MSHIFT=$ms
    echo "$l $ent | $(python3 tools/small_sums_inflight.py $l 2>&1 | tail -n 1)"
  done
done
