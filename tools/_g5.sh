mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_phase_parity.py tests/test_gpu_sweep.py -m gpu -q -x > gpurun_out/r6/t5_e2e.log 2>&1; echo "rc e2e $?" >> gpurun_out/r6/t5_e2e.log
for cfg in c4 c5s; do
  python bench.py --config $cfg --steps 20 --warmup 3 --no-api --no-cpu-baseline > gpurun_out/r6/t5_bench_${cfg}_new.json 2> gpurun_out/r6/t5_bench_${cfg}_new.err
  MCL_NO_ROW_PREFETCH=1 python bench.py --config $cfg --steps 20 --warmup 3 --no-api --no-cpu-baseline > gpurun_out/r6/t5_bench_${cfg}_old.json 2> gpurun_out/r6/t5_bench_${cfg}_old.err
done
python bench.py --config c5 --steps 5 --warmup 30 --no-api --no-cpu-baseline > gpurun_out/r6/t5_bench_c5_new.json 2> gpurun_out/r6/t5_bench_c5_new.err
MCL_NO_ROW_PREFETCH=1 python bench.py --config c5 --steps 5 --warmup 30 --no-api --no-cpu-baseline > gpurun_out/r6/t5_bench_c5_old.json 2> gpurun_out/r6/t5_bench_c5_old.err
tail -n 3 gpurun_out/r6/t5_e2e.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6/t5_bench_*.json')):
    try:
        d=json.load(open(f)); ch=[e for e in d['roofline']['per_kernel'] if 'chained' in e['role']]
        print(f.split('t5_bench_')[1], d['value'], d['ms_per_step'], [(e['kernel'][:40], e['launches_per_step'], e['avg_us'], e['frac']) for e in ch])
    except Exception as e: print(f, 'ERR', e)
PY
