for d in 0 15; do
  echo "== dbg $d"
  MCL_SWEEP_DBG=$d timeout 200 python bench.py --config c3 --steps 80000 --warmup 5 > /tmp/b_$d.log 2>&1 &
  PID=$!
  sleep 12
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction" | head -4 | tr '\n' ' '; echo; sleep 1; done
  wait $PID
  tail -1 /tmp/b_$d.log | cut -c1-120
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
