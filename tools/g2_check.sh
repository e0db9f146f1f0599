#!/bin/bash
# G2 accumulation in three contexts: the G2 sum alone, as part of bench.py's extras, inside PHGR13Prove.
python3 bench.py --group g2 --no-extras --no-cpu-baseline --steps 12 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('g2 alone', round(d['ms_per_step'],3), round(d['ms_per_step_one_at_a_time'],3))"
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']; print('g1', round(d['ms_per_step'],3), 'extras g2', round(e['g2_msm_2p20']['ms_per_step'],3), 'g16', round(e['groth16_prove_2p20']['ms'],2), 'phgr13', round(e['phgr13_prove_2p20']['ms'],2))"
REPS=3 python3 tools/phgr13_experiment.py 2>/dev/null | tail -1
