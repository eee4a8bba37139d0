mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -x -k "scale or config5 or config4" > gpurun_out/r6/t12_e2e.log 2>&1; echo "rc $?" >> gpurun_out/r6/t12_e2e.log; tail -3 gpurun_out/r6/t12_e2e.log
grep -q "rc 0" gpurun_out/r6/t12_e2e.log || exit 1
timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 30 --no-api --no-cpu-baseline > gpurun_out/r6/t12_bench_c5_new.json 2> gpurun_out/r6/t12_bench_c5_new.err &&
MCL_NO_XC_LDS=1 timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 30 --no-api --no-cpu-baseline > gpurun_out/r6/t12_bench_c5_old.json 2> gpurun_out/r6/t12_bench_c5_old.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/t12_bench_*.json')):
    try:
        d=json.load(open(f)); ch=[e for e in d['roofline']['per_kernel'] if 'X C' in e['role']]
        print(f.split('t12_bench_')[1], d['value'], d['ms_per_step'], [(e['kernel'][:40], e['launches_per_step'], e['avg_us'], e['frac']) for e in ch])
    except Exception as e: print(f, 'ERR', e)
PY
