set -x
mkdir -p gpurun_out/r4a
rocprofv3 -L > gpurun_out/r4a/counters.txt 2>&1 || true
tools/ntt_ab.sh main batch batch8 tune8:6:9 tune8:7:9 tune8:6:10 tune:9:10 > gpurun_out/r4a/ntt_ab.txt 2>&1
tail -20 gpurun_out/r4a/ntt_ab.txt
tools/sq_pmc.sh gpurun_out/r4a/sq_bench.json bench
tools/sq_pmc.sh gpurun_out/r4a/sq_g16.json g16
python3 bench.py --no-extras --no-cpu-baseline > gpurun_out/r4a/bench_quick.json 2> gpurun_out/r4a/bench_quick.err
tail -c 600 gpurun_out/r4a/bench_quick.json
