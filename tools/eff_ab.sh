#!/bin/bash
# A/B on one box: the shipped library against playsnark_amd/libps_<name>.so, both provers at 2^20 (toy circuit, booleanity gates)
for round in 1 2; do for name in main "$@"; do
  if [ $name = main ]; then unset PLAYSNARK_HIP_LIB; else export PLAYSNARK_HIP_LIB=$PWD/playsnark_amd/libps_$name.so; fi
  echo "$name | g16 $(REPS=5 python3 tools/g16_experiment.py 2>&1 | grep 'groth16 ms' | sed 's/{.*//') | g16 bits $(CIRCUIT=bits REPS=5 python3 tools/g16_experiment.py 2>&1 | grep 'groth16 ms' | sed 's/{.*//') | phgr13 $(REPS=5 python3 tools/phgr13_experiment.py 2>&1 | grep 'phgr13 ms' | sed 's/{.*//')"
done; done
