#!/bin/bash
# The side measurements DESIGN.md quotes next to the bench line: plain plan (no window table), witness regime, G2.
run() { python3 bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*', '| pipelined %.3f one-at-a-time %.3f h2d %.3f | c=%d W=%d | acc %.3f' % (d['ms_per_step'], d['ms_per_step_one_at_a_time'], d['ms_per_step_with_h2d'], c['window_bits'], c['windows'], d['roofline']['kernel_ms_one_at_a_time']))"; }
run
run --no-table
run --scalars witness
run --scalars witness --no-table
run --group g2
run --group g2 --no-table
run --log2n 16
run --log2n 22 --steps 8
