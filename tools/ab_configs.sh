for c in k512 r32 c3_8th; do
  for m in 1 0; do
    if [ $m = 1 ]; then export MCL_NO_SWEEP=1; else unset MCL_NO_SWEEP; fi
    python bench.py --config $c --steps 30 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$c', 'two-pass' if $m else 'sweep', d['value'], d['ms_per_step'], d['roofline']['all_kernels_avg_us'] if d['roofline'] else None)"
  done
done
