#!/bin/bash
# Issue / stall counters of the hot kernels (rocprofv3 --pmc, SQ block: 8 slots per pass; counters alone, no trace domains).
#   tools/sq_pmc.sh <out.json> bench|g16    (GPU box, repository root)
# bench: python3 bench.py one sum at a time (k_accumulate alone on the chip); g16: Groth16Prove at 2^20 with the reference's
# key form (k_ntt_pass).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X guide).
set -e
out=$1; what=$2
export TMPDIR=/tmp
passes=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
)
dbs=()
i=0
for p in "${passes[@]}"; do
  d=$(mktemp -d /tmp/sq.XXXX)
  if [ "$what" = bench ]; then
    rocprofv3 --pmc $p -d "$d" -o run -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --in-flight 1 > /dev/null 2> "$out.err$i" || { echo "pass $i failed"; tail -3 "$out.err$i"; i=$((i+1)); continue; }
  else
    MONOMIAL=1 REPS=2 rocprofv3 --pmc $p -d "$d" -o run -- python3 tools/g16_experiment.py > /dev/null 2> "$out.err$i" || { echo "pass $i failed"; tail -3 "$out.err$i"; i=$((i+1)); continue; }
  fi
  cp "$(find "$d" -name '*.db' | head -1)" "/tmp/sq_${what}_$i.db"
  dbs+=("/tmp/sq_${what}_$i.db")
  rm -rf "$d"
  i=$((i+1))
done
python3 tools/rocpd_pmc.py "${dbs[@]}" > "$out"
