#!/bin/bash
# Round profiles on the GPU box: bench line, kernel statistics (rocprofv3 --kernel-trace --stats) of the bench and of
# both provers at 2^20, PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, as the MI355X guide prescribes) for the
# accumulation and for the NTT passes.  Usage (repository root): tools/collect_profiles.sh <out-dir> <tag>
set -e
out=$1; tag=$2
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
echo "bench done"
tools/prof_bench.sh "$out/${tag}_kernel_stats_bench_msm_2p20" --steps 20 --warmup 5 --no-extras --no-cpu-baseline
tools/prof_bench.sh "$out/${tag}_kernel_stats_bench_msm_2p20_one_at_a_time" --steps 20 --warmup 5 --no-extras --no-cpu-baseline --in-flight 1
echo "kernel stats of the bench done"
for what in g16 phgr13; do
  d=$(mktemp -d /tmp/prof.XXXX)
  REPS=5 rocprofv3 --kernel-trace --stats -d "$d" -o run -- python3 tools/${what}_experiment.py > "$out/${tag}_${what}_2p20.log" 2>/dev/null
  python3 tools/rocpd_stats.py "$(find "$d" -name '*.db' | head -1)" > "$out/${tag}_kernel_stats_${what}_2p20.csv"
  rm -rf "$d"
done
echo "kernel stats of the provers done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  rocprofv3 --pmc $ctr -d "$d" -o run -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --in-flight 1 > /dev/null 2>&1
  cp "$(find "$d" -name '*.db' | head -1)" "/tmp/pmc_bench_$ctr.db"
  rm -rf "$d"
  d=$(mktemp -d /tmp/pmc.XXXX)
  REPS=2 rocprofv3 --pmc $ctr -d "$d" -o run -- python3 tools/g16_experiment.py > /dev/null 2>&1
  cp "$(find "$d" -name '*.db' | head -1)" "/tmp/pmc_g16_$ctr.db"
  rm -rf "$d"
done
python3 tools/rocpd_pmc.py /tmp/pmc_bench_FETCH_SIZE.db /tmp/pmc_bench_WRITE_SIZE.db > "$out/${tag}_pmc_bench_msm_2p20.json"
# the same two passes for the G2 accumulation (bench.py --group g2 carries their sum as roofline.traffic)
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$(mktemp -d /tmp/pmc.XXXX)
  rocprofv3 --pmc $ctr -d "$d" -o run -- python3 bench.py --group g2 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --in-flight 1 > /dev/null 2>&1
  cp "$(find "$d" -name '*.db' | head -1)" "/tmp/pmc_g2_$ctr.db"
  rm -rf "$d"
done
python3 tools/rocpd_pmc.py /tmp/pmc_g2_FETCH_SIZE.db /tmp/pmc_g2_WRITE_SIZE.db > "$out/${tag}_pmc_bench_msm_2p20_g2.json"
python3 tools/rocpd_pmc.py /tmp/pmc_g16_FETCH_SIZE.db /tmp/pmc_g16_WRITE_SIZE.db > "$out/${tag}_pmc_groth16_2p20.json"
echo "pmc passes done"
# round 4: issue / stall counters of the hot kernels, the quotient phase kernel by kernel, the shares of a sharded proof
tools/sq_pmc.sh "$out/${tag}_sq_counters_bench_one_at_a_time.json" bench
d=$(mktemp -d /tmp/prof.XXXX)
MONOMIAL=1 REPS=3 rocprofv3 --kernel-trace -d "$d" -o run -- python3 tools/g16_experiment.py > /dev/null 2>&1
python3 tools/quotient_breakdown.py "$(find "$d" -name '*.db' | head -1)" > "$out/${tag}_quotient_breakdown.txt"
rm -rf "$d"
python3 tools/g16_shares.py > "$out/${tag}_g16_shares.txt" 2>&1
echo "round-4 extras done"
