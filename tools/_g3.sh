mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_condition_probe.py tests/test_gpu_cabi_contract.py -m gpu -q -s > gpurun_out/r6/t3_new.log 2>&1; echo "rc new $?" >> gpurun_out/r6/t3_new.log
python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s -k "scale" > gpurun_out/r6/t3_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r6/t3_e2e.log
MCL_FUZZ_REPORT_ONLY=1 MCL_FUZZ_MID_SEEDS=400 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "mid_size or larger_problem" > gpurun_out/r6/t3_mid400.log 2>&1; echo "rc $?" >> gpurun_out/r6/t3_mid400.log
python bench.py --config c3 --steps 20 --warmup 3 > gpurun_out/r6/t3_bench_c3.json 2> gpurun_out/r6/t3_bench_c3.err
python bench.py --config c4 --steps 20 --warmup 3 > gpurun_out/r6/t3_bench_c4.json 2> gpurun_out/r6/t3_bench_c4.err
python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -x > gpurun_out/r6/t3_contract.log 2>&1; echo "rc $?" >> gpurun_out/r6/t3_contract.log
tail -n 3 gpurun_out/r6/t3_new.log gpurun_out/r6/t3_e2e.log gpurun_out/r6/t3_mid400.log gpurun_out/r6/t3_contract.log; cut -c1-200 gpurun_out/r6/t3_bench_c3.json gpurun_out/r6/t3_bench_c4.json
