#!/bin/bash
# Quotient A/B on ONE box, no profiler (round 3: the shipped pass kernel against -DPS_NTT_PASS8 builds): libraries playsnark_amd/libps_<name>.so (and "main" = the shipped one), optional
# PS_NTT_MAXK / PS_NTT_TILE per entry ("name:maxk:tile", measurement builds with -DPS_NTT_TUNE only).  Prints the Groth16Prove phases at 2^20
# constraints with the reference's key form (monomial: the quotient is interpolations + division) and with the Lagrange
# form (values route), two rounds interleaved.
for round in 1 2; do
  for ent in "$@"; do
    IFS=: read -r name maxk tile <<< "$ent"
    if [ "$name" = main ]; then unset PLAYSNARK_HIP_LIB; else export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libps_$name.so; fi
    [ -n "$maxk" ] && export PS_NTT_MAXK=$maxk || unset PS_NTT_MAXK
    [ -n "$tile" ] && export PS_NTT_TILE=$tile || unset PS_NTT_TILE
    m=$(MONOMIAL=1 REPS=4 python3 tools/g16_experiment.py 2>/dev/null | grep "groth16 ms" | sed "s/.*quotient': \([0-9.]*\).*total': \([0-9.]*\).*/monomial: quotient \1 total \2/")
    l=$(REPS=4 python3 tools/g16_experiment.py 2>/dev/null | grep "groth16 ms" | sed "s/.*quotient': \([0-9.]*\).*total': \([0-9.]*\).*/values: quotient \1 total \2/")
    echo "$ent | $m | $l"
  done
done
