#!/bin/bash
# The first run on a node with more than one MI355X (docs/multi_gpu_bringup.md as a script; VERDICT r5 #5c).
#   bash tools/first_multi_gpu_run.sh [out_dir=gpurun_out/multi_gpu] [max_gpus=8]
# Emits the BASELINE metric lines at 1 / 2 / 4 / 8 devices for configs 3, 4 and 5 - one JSON line per (config, N) in
# <out_dir>/<config>_n<N>.json, stderr beside it - plus the evidence that the collectives really crossed devices:
#   * NCCL_DEBUG=INFO log of the first N > 1 run (<out_dir>/rccl_init_n2.log: "comm ... nranks 2", the ring / tree topology),
#   * every N > 1 line's own guards: collective_path, collectives_per_step, replicated_C_bit_identical,
#   * final_rel_rec_error of every N against the N = 1 line of the same config (equal to ~1e-6),
# and a summary table with the scaling efficiency computed from the lines.  Stops at the first failing step and says which
# switch isolates it.  Nothing here needs the network; the rendezvous is 127.0.0.1.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/multi_gpu}; MAXN=${2:-8}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NDEV (asked for up to $MAXN)"
if [ "$NDEV" -lt 2 ]; then echo "fewer than two devices: nothing to bring up here (single-device numbers: python bench.py)"; exit 2; fi

run() {  # run <config> <n> <steps> <warmup> [env ...]
  local cfg=$1 n=$2 steps=$3 warm=$4; shift 4
  local tag="$OUT/${cfg}_n${n}"
  echo "== config $cfg on $n device(s)"
  if ! env "$@" timeout -k 10 900 python3 bench.py --config "$cfg" --gpus "$n" --steps "$steps" --warmup "$warm" --no-api \
        > "$tag.json" 2> "$tag.err"; then
    echo "FAILED: $cfg at N=$n (stderr: $tag.err; the last '[bench] <leg>' line names the leg that hung)"
    echo "  isolate: MCL_NO_DIRECT_RCCL=1 (collectives through torch.distributed), MCL_RCCL_INIT_TIMEOUT_S=120 (slow ncclCommInitRank),"
    echo "           NCCL_DEBUG=INFO (RCCL's topology log)"
    tail -5 "$tag.err"
    return 1
  fi
  python3 - "$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"   {d['value']:.2f} {d['unit']}  ({d['ms_per_step']:.4f} ms/step)  path: {d.get('collective_path', 'single device')}  "
      f"collectives/step {d.get('collectives_per_step')}  C bit-identical {d.get('replicated_C_bit_identical')}  "
      f"rec {d.get('final_rel_rec_error')}")
if d["n_gpus"] > 1 and d.get("replicated_C_bit_identical") is not True:
    sys.exit("   the replicated C differs between the ranks: the all-reduce did not hand every rank the same bits")
PY
}

NS="1"; for n in 2 4 8; do if [ "$n" -le "$NDEV" ] && [ "$n" -le "$MAXN" ]; then NS="$NS $n"; fi; done
# 1. the metric configuration; the first N > 1 run with RCCL's own log as evidence that the ranks found each other
run c3 1 200 20 || exit 1
run c3 2 200 20 NCCL_DEBUG=INFO NCCL_DEBUG_FILE="$OUT/rccl_init_n2.log" || exit 1
grep -h -m 3 -i "nranks\|Ring\|Connected" "$OUT"/rccl_init_n2.log* 2>/dev/null | cut -c1-200 || echo "   (no RCCL log written: check NCCL_DEBUG_FILE support)"
for n in $NS; do if [ "$n" -gt 2 ]; then run c3 "$n" 200 20 || exit 1; fi; done
# 2. config 4: six collectives per step (PARAFAC2: one r*r + 1 all-reduce per inner iteration), the step API
for n in $NS; do run c4 "$n" 50 10 || exit 1; done
# 3. config 5: the configuration sized for the node (X = 68.7 GB); steady state of the unimodal regressions needs ~25 iterations
for n in $NS; do run c5 "$n" 4 30 || exit 1; done
# 4. the public API under real RCCL
python3 -m pytest tests/test_gpu_api_surface.py -k "rccl or sharing" -q > "$OUT/api_surface.log" 2>&1 && echo "== cmf_aoadmm(group=) over RCCL: passed" \
  || { echo "== cmf_aoadmm(group=) over RCCL: FAILED ($OUT/api_surface.log)"; tail -5 "$OUT/api_surface.log"; }

python3 - "$OUT" <<'PY'
import glob, json, os, sys
rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "c*_n*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    rows.setdefault(os.path.basename(f).split("_n")[0], {})[d["n_gpus"]] = d
print("\nconfig   N   it/s        speed-up  efficiency  rec error vs N=1")
for cfg in sorted(rows):
    base = rows[cfg].get(1)
    for n in sorted(rows[cfg]):
        d = rows[cfg][n]
        sp = d["value"] / base["value"] if base else float("nan")
        dr = abs(d.get("final_rel_rec_error", 0) - base.get("final_rel_rec_error", 0)) if base else float("nan")
        print(f"{cfg:7s} {n:2d}  {d['value']:10.2f}  {sp:7.2f}   {sp / n:8.2f}   {dr:.1e}")
PY
