#!/bin/bash
# Both provers per circuit size (Lagrange-form key, and the monomial key for Groth16): ms per proof and its phases.
for l in "$@"; do
  echo "n = 2^$l"
  LOG2N=$l REPS=3 python3 tools/g16_experiment.py 2>/dev/null | grep -E "groth16 ms|device setup"
  LOG2N=$l REPS=3 MONOMIAL=1 python3 tools/g16_experiment.py 2>/dev/null | grep "groth16 ms" | sed 's/^/  monomial key:/'
  LOG2N=$l REPS=3 python3 tools/phgr13_experiment.py 2>/dev/null | grep "phgr13 ms"
done
