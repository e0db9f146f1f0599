#!/bin/bash
# A/B of library builds on ONE box (box-to-box spread is ~5 %): tools/ab_libs.sh <name> ... runs, for every
# playsnark_amd/libps_<name>.so given, the G1 / G2 bench legs and both provers, twice, interleaved.
for round in 1 2; do
  for v in "$@"; do
    export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libps_$v.so
    g1=$(python3 bench.py --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g1 %.3f / %.3f' % (d['ms_per_step'], d['ms_per_step_one_at_a_time']))")
    g2=$(python3 bench.py --group g2 --no-extras --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2 %.3f / %.3f' % (d['ms_per_step'], d['ms_per_step_one_at_a_time']))")
    ph=$(REPS=4 python3 tools/phgr13_experiment.py 2>/dev/null | grep "phgr13 ms" | sed 's/{.*//')
    g16=$(REPS=4 python3 tools/g16_experiment.py 2>/dev/null | grep "groth16 ms" | sed 's/{.*//')
    echo "$v | $g1 | $g2 | $ph | $g16"
  done
done
