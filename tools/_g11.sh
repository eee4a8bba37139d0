mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_phase_parity.py tests/test_more_penalties.py tests/test_gpu_fuzz_parity.py -m gpu -q > gpurun_out/r6/t11_tests.log 2>&1; echo "rc $?" >> gpurun_out/r6/t11_tests.log; tail -3 gpurun_out/r6/t11_tests.log
python tools/exact_mode_cost_mid.py > gpurun_out/r6/t11_exact_cost.log 2>&1; grep "us/iter" gpurun_out/r6/t11_exact_cost.log
bash tools/evidence_round.sh 6 c3 20 sq 3 > gpurun_out/r6/t11_ev_c3.log 2>&1; tail -2 gpurun_out/r6/t11_ev_c3.log
bash tools/evidence_round.sh 6 c4 20 nosq 3 > gpurun_out/r6/t11_ev_c4.log 2>&1; tail -2 gpurun_out/r6/t11_ev_c4.log
bash tools/evidence_round.sh 6 c2 20 nosq 3 > gpurun_out/r6/t11_ev_c2.log 2>&1; tail -2 gpurun_out/r6/t11_ev_c2.log
