# final round-1 evidence: kernel-trace stats and PMC traffic (separate passes) for config 3 (one-pass sweep), its two-pass
# reference (MCL_NO_SWEEP=1) and the K = 512 variant
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o c3 -- python3 /root/repo/bench.py --config c3 --steps 50 --warmup 5 > $O.c3.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k512 -o k512 -- python3 /root/repo/bench.py --config k512 --steps 50 --warmup 5 > $O.k512.json 2>/dev/null
export MCL_NO_SWEEP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_twopass -o c3 -- python3 /root/repo/bench.py --config c3 --steps 50 --warmup 5 > $O.c3_twopass.json 2>/dev/null
unset MCL_NO_SWEEP
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c3_fetch -o f -- python3 /root/repo/bench.py --config c3 --steps 5 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c3_write -o w -- python3 /root/repo/bench.py --config c3 --steps 5 --warmup 2 > /dev/null 2>&1
tail -n1 $O.c3.json | cut -c1-300
tail -n1 $O.k512.json | cut -c1-300
tail -n1 $O.c3_twopass.json | cut -c1-300
ls $O/*
