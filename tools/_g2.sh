mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_condition_probe.py tests/test_gpu_closed_form.py tests/test_gpu_external_penalties.py tests/test_gpu_cabi_contract.py tests/test_more_penalties.py -m gpu -q -s > gpurun_out/r6/t2_new.log 2>&1; echo "rc new $?" >> gpurun_out/r6/t2_new.log
python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s -k "golden or public or scale or degenerate" > gpurun_out/r6/t2_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r6/t2_e2e.log
MCL_FUZZ_REPORT_ONLY=1 MCL_FUZZ_MID_SEEDS=400 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -s -k "mid_size or larger_problem" > gpurun_out/r6/t2_mid400.log 2>&1; echo "rc $?" >> gpurun_out/r6/t2_mid400.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/prof_exact_c4 -o exact_c4 -- python3 $GRAFT_REPO_ROOT/tools/exact_stack_prof.py c4 48 576 256 16 50 > $GRAFT_REPO_ROOT/gpurun_out/r6/t2_prof_c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/prof_exact_c3 -o exact_c3 -- python3 $GRAFT_REPO_ROOT/tools/exact_stack_prof.py c3 64 512 256 16 50 > $GRAFT_REPO_ROOT/gpurun_out/r6/t2_prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 3 gpurun_out/r6/t2_new.log gpurun_out/r6/t2_e2e.log gpurun_out/r6/t2_mid400.log gpurun_out/r6/t2_prof_c4.log
