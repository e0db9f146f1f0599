set -x
mkdir -p gpurun_out/r4b
python3 -m pytest tests/test_msm_gpu.py -x -q -k "views or window_table" > gpurun_out/r4b/pytest_views.log 2>&1; tail -3 gpurun_out/r4b/pytest_views.log
python3 -m pytest tests/test_bench_host.py -x -q -m gpu > gpurun_out/r4b/pytest_bench.log 2>&1; tail -3 gpurun_out/r4b/pytest_bench.log
python3 -m pytest tests/test_scale_gpu.py -x -q -m gpu > gpurun_out/r4b/pytest_scale.log 2>&1; tail -3 gpurun_out/r4b/pytest_scale.log
python3 tools/g16_shares.py > gpurun_out/r4b/g16_shares.txt 2>&1; tail -12 gpurun_out/r4b/g16_shares.txt
python3 tools/slice_ab.py g1 20 24,26,28,30,32,40,52 > gpurun_out/r4b/slice_ab.txt 2>&1; cat gpurun_out/r4b/slice_ab.txt
cp playsnark_amd/libplaysnark_hip.so playsnark_amd/libps_main.so
tools/ab_libs.sh main nochain > gpurun_out/r4b/ab_nochain.txt 2>&1; cat gpurun_out/r4b/ab_nochain.txt
