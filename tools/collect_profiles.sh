# copy the judged summaries of tools/evidence.sh (gpurun_out/ev, scratch) into profiles/ (tracked); run from the repo root
E=gpurun_out/ev
cp $E/c3/c3_kernel_stats.csv profiles/r1_c3_kernel_stats.csv
cp $E/c3_two_pass/c3_two_pass_kernel_stats.csv profiles/r1_c3_two_pass_kernel_stats.csv
cp $E/k512/k512_kernel_stats.csv profiles/r1_k512_kernel_stats.csv
cp $E/c4/c4_kernel_stats.csv profiles/r1_c4_kernel_stats.csv
cp $E/c3_8th/c3_8th_kernel_stats.csv profiles/r1_c3_eighth_shard_kernel_stats.csv
cp $E/c5full/c5full_kernel_stats.csv profiles/r1_c5_fullsize_kernel_stats.csv
python3 - <<'PY'
import csv
# keep only the FETCH_SIZE / WRITE_SIZE rows of the library's kernels (the raw counter CSVs are large)
for tag, src in (("FETCH_SIZE", "gpurun_out/ev/c3_fetch/f_counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/ev/c3_write/w_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(src)) if r["Counter_Name"] == tag and "k_" in r["Kernel_Name"]]
    with open(f"profiles/r1_c3_pmc_{tag}.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
PY
python3 tools/pmc_traffic.py profiles/r1_c3_pmc_FETCH_SIZE.csv profiles/r1_c3_pmc_WRITE_SIZE.csv profiles/r1_c3_pmc_traffic.json "config 3 (one-pass sweep), rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of bench.py --config c3 --steps 5 --warmup 2; FETCH_SIZE doubled (gfx950 correction), KiB -> bytes" > /dev/null
for n in c3 c4; do tail -n1 $E/$n.json > profiles/r1_${n}_bench_under_rocprof.json; done
tail -n1 $E/bench_c3_plain.json > profiles/r1_c3_bench.json
grep '"config"' $E/c5full.json > profiles/r1_c5_fullsize_run.json
