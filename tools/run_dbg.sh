for d in 0 15; do echo "dbg $d"; MCL_SWEEP_DBG=$d python bench.py --config c3 --steps 30 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['all_kernels_avg_us'] if d['roofline'] else None)"; done
