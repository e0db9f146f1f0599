#!/bin/bash
# One MSM per length 2^10 .. 2^23 (G1; 2^10 .. 2^21 for G2), with and without the window table: ms per sum (four in
# flight / alone), window, slice, ns per point -- to spot lengths where a plan choice goes wrong (2^19 once did).
for g in g1 g2; do
  hi=23; [ $g = g2 ] && hi=21
  for l in $(seq 10 $hi); do
    for t in "" "--no-table"; do
      python3 bench.py --group $g --log2n $l $t --no-extras --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; n=c['points_per_gpu']
print('$g 2^%-2d %-9s c=%-2d W=%-2d M=%-4d  %8.3f ms in flight (%6.2f ns/pt)  %8.3f alone' % ($l, 'table' if c['window_table'] else 'plain', c['window_bits'], c['windows'], c['slice'], d['ms_per_step'], d['ms_per_step']*1e6/n, d['ms_per_step_one_at_a_time']))"
    done
  done
done
