mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_condition_probe.py tests/test_gpu_unimodal_kernels.py -m gpu -q -s > gpurun_out/r6/t4_new.log 2>&1; echo "rc new $?" >> gpurun_out/r6/t4_new.log
python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s -k "scale or config5" > gpurun_out/r6/t4_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r6/t4_e2e.log
MCL_FUZZ_REPORT_ONLY=1 MCL_FUZZ_MID_SEEDS=400 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "mid_size or larger_problem" > gpurun_out/r6/t4_mid400.log 2>&1; echo "rc $?" >> gpurun_out/r6/t4_mid400.log
python tools/uni_lib_ab.py build_ab/uni_old.so matcouply_amd/libmatcouply_hip.so c5 30 > gpurun_out/r6/t4_uni_ab.log 2>&1; echo "rc $?" >> gpurun_out/r6/t4_uni_ab.log
tail -n 3 gpurun_out/r6/t4_new.log gpurun_out/r6/t4_e2e.log gpurun_out/r6/t4_mid400.log; tail -n 12 gpurun_out/r6/t4_uni_ab.log
