mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -q -x > gpurun_out/r6/t10_full.log 2>&1; echo "rc $?" >> gpurun_out/r6/t10_full.log; tail -6 gpurun_out/r6/t10_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/t10_smoke.log 2>&1; tail -2 gpurun_out/r6/t10_smoke.log
