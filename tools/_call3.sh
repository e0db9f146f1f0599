set -x
mkdir -p gpurun_out/r4c
python3 -m pytest tests/test_prover_gpu.py tests/test_scale_gpu.py -x -q -m gpu > gpurun_out/r4c/pytest_prover.log 2>&1; tail -3 gpurun_out/r4c/pytest_prover.log
python3 -m pytest tests/test_msm_gpu.py -x -q -k "views or window_table or multi" > gpurun_out/r4c/pytest_views.log 2>&1; tail -3 gpurun_out/r4c/pytest_views.log
tools/ntt_ab.sh main batch > gpurun_out/r4c/ntt_ab.txt 2>&1; cat gpurun_out/r4c/ntt_ab.txt
python3 tools/g16_shares.py > gpurun_out/r4c/g16_shares.txt 2>&1; tail -9 gpurun_out/r4c/g16_shares.txt
cp playsnark_amd/libplaysnark_hip.so playsnark_amd/libps_main.so
tools/ab_libs.sh main batch > gpurun_out/r4c/ab.txt 2>&1; cat gpurun_out/r4c/ab.txt
