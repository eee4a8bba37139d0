# A/B of two builds of the library on the same box: usage ab_lib.sh <old.so> <new.so> [config] [steps]
cfg=${3:-c3}; steps=${4:-50}
cp matcouply_amd/libmatcouply_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for which in "$1" "$2"; do
    cp "$which" matcouply_amd/libmatcouply_hip.so
    python bench.py --config $cfg --steps $steps --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$which', d['value'], d['ms_per_step'], d['roofline']['all_kernels_avg_us'])"
  done
done
cp /tmp/lib_keep.so matcouply_amd/libmatcouply_hip.so
