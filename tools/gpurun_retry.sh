#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'   - retries ONLY when gpurun reports "no box / slot free" (exit 3, nothing charged)
t=$1; shift
for attempt in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] no slot (attempt $attempt), sleeping 120 s"
  sleep 120
done
exit 3
