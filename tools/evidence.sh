# Round evidence set on one GPU box (run via gpurun from the repo root): kernel-trace stats for config 3 (sweep), its two-pass
# reference, K = 512, config 4, the 1/8 shard of config 3, config 5 at full size; PMC traffic passes for config 3.
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/ev; mkdir -p $O
prof() { # name, command...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o $n -- "$@" > $O/$n.json 2> $O/$n.err || return 1
  tail -n1 $O/$n.json | cut -c1-160
}
prof c3 python3 $R/bench.py --config c3 --steps 50 --warmup 5 || exit 1
prof k512 python3 $R/bench.py --config k512 --steps 50 --warmup 5 --no-cpu-baseline || exit 1
MCL_NO_SWEEP=1 prof c3_two_pass python3 $R/bench.py --config c3 --steps 50 --warmup 5 --no-cpu-baseline || exit 1
prof c4 python3 $R/bench.py --config c4 --steps 30 --warmup 3 || exit 1
prof c3_8th python3 $R/bench.py --config c3_8th --steps 200 --warmup 10 --no-cpu-baseline || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -o f -- python3 $R/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_write -o w -- python3 $R/bench.py --config c3 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5full -o c5full -- python3 $R/tools/run_c5_full.py > $O/c5full.json 2> $O/c5full.err || exit 1
grep '"config"' $O/c5full.json | cut -c1-200
python3 $R/bench.py --config c3 --steps 200 --warmup 10 > $O/bench_c3_plain.json 2>/dev/null; tail -n1 $O/bench_c3_plain.json | cut -c1-160
ls $O
