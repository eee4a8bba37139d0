mkdir -p gpurun_out/r6
timeout -k 10 600 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -x -k "scale or config5" > gpurun_out/r6/t13_e2e.log 2>&1; echo "rc $?" >> gpurun_out/r6/t13_e2e.log; tail -3 gpurun_out/r6/t13_e2e.log
grep -q "rc 0" gpurun_out/r6/t13_e2e.log || exit 1
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --config c5 --steps 5 --warmup 30 --no-api --no-cpu-baseline > gpurun_out/r6/t13_bench_c5_$tag.json 2> gpurun_out/r6/t13_bench_c5_$tag.err; }
run d8 MCL_XC_LDS_DEPTH=8 && run d4 MCL_XC_LDS_DEPTH=4 && run d8xt8 MCL_XC_LDS_DEPTH=8 MCL_XT_DEPTH=8 && run old MCL_NO_XC_LDS=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/t13_bench_*.json')):
    try:
        d=json.load(open(f)); ch=[e for e in d['roofline']['per_kernel'] if 'X' in e['role'][:3]]
        print(f.split('t13_bench_')[1], d['value'], d['ms_per_step'], [(e['kernel'][:34], e['avg_us'], e['frac']) for e in ch])
    except Exception as e: print(f, 'ERR', e)
PY
