mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_phase_parity.py tests/test_gpu_fuzz_parity.py tests/test_more_penalties.py tests/test_gpu_condition_probe.py -m gpu -q > gpurun_out/r6/t7_tests.log 2>&1; echo "rc $?" >> gpurun_out/r6/t7_tests.log; tail -4 gpurun_out/r6/t7_tests.log
python tools/exact_mode_cost_mid.py > gpurun_out/r6/t7_exact_cost.log 2>&1; grep "^c4" gpurun_out/r6/t7_exact_cost.log
