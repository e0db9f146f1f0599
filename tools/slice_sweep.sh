#!/bin/bash
# Accumulation slice length M (sorted entries per thread) against the 2^20 G1 sum: pipelined and one at a time.
for m in "$@"; do
  python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --slice $m 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M=$m', 'pipelined %.3f' % d['ms_per_step'], 'one-at-a-time %.3f' % d['ms_per_step_one_at_a_time'], 'acc %.3f' % d['stage_ms']['accumulate'], 'fixup %.3f' % d['stage_ms']['fixup'])"
done
