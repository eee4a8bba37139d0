mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_closed_form.py tests/test_gpu_external_penalties.py tests/test_more_penalties.py -m gpu -q -x -s > gpurun_out/r6/t1_new.log 2>&1; echo "rc new $?" >> gpurun_out/r6/t1_new.log
python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s -k "golden or public or scale or degenerate" > gpurun_out/r6/t1_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r6/t1_e2e.log
MCL_FUZZ_REPORT_ONLY=1 MCL_FUZZ_MID_SEEDS=120 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "mid_size" > gpurun_out/r6/t1_mid_default.log 2>&1; echo "rc $?" >> gpurun_out/r6/t1_mid_default.log
MCL_EXACT=1 MCL_FUZZ_REPORT_ONLY=1 MCL_FUZZ_MID_SEEDS=120 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "mid_size" > gpurun_out/r6/t1_mid_exact.log 2>&1; echo "rc $?" >> gpurun_out/r6/t1_mid_exact.log
python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "known_outside or residue" > gpurun_out/r6/t1_outside.log 2>&1; echo "rc $?" >> gpurun_out/r6/t1_outside.log
python tools/exact_mode_cost_mid.py > gpurun_out/r6/t1_exact_cost.log 2>&1
tail -3 gpurun_out/r6/t1_new.log gpurun_out/r6/t1_e2e.log gpurun_out/r6/t1_mid_default.log gpurun_out/r6/t1_mid_exact.log gpurun_out/r6/t1_outside.log
