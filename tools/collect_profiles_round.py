"""gpurun_out/ev<round>/<config>/ (tools/evidence_round.sh) -> profiles/r<round>_<config>_{bench.json,
bench_under_rocprof.json, kernel_stats.csv, pmc_traffic.json[, sq_counters.json]}.  Run from the repo root after the gpurun
call:  python tools/collect_profiles_round.py <round> [config ...]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1]
EV, PR = os.path.join(REPO, "gpurun_out", f"ev{RND}"), os.path.join(REPO, "profiles")


def short(name):
    return re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]


def counters(path):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


for cfg in (sys.argv[2:] or sorted(os.listdir(EV))):
    d = os.path.join(EV, cfg)
    shutil.copy(os.path.join(d, "bench.json"), os.path.join(PR, f"r{RND}_{cfg}_bench.json"))
    shutil.copy(os.path.join(d, "bench_traced.json"), os.path.join(PR, f"r{RND}_{cfg}_bench_under_rocprof.json"))
    stats = glob.glob(os.path.join(d, "trace", "*kernel_stats.csv"))
    rows = [r for r in csv.DictReader(open(stats[0])) if short(r["Name"]).startswith("k_")]
    with open(os.path.join(PR, f"r{RND}_{cfg}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
    dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}
    fe = counters(glob.glob(os.path.join(d, "fetch", "*counter_collection.csv"))[0])
    wr = counters(glob.glob(os.path.join(d, "write", "*counter_collection.csv"))[0])
    out = {"note": f"bench.py --config {cfg} under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE "
                   "doubled (gfx950 reports 1/2 of a wide streaming read, MI355X_MICROARCH.md), KiB -> bytes; WRITE_SIZE exact; "
                   "avg_us from the kernel-trace pass of the same command", "kernels": {}}
    for (name, cn), (n, v) in fe.items():
        if cn != "FETCH_SIZE":
            continue
        wn, wv = wr.get((name, "WRITE_SIZE"), [n, 0.0])
        fb, wb = 2 * 1024 * v / n, 1024 * wv / max(wn, 1)
        e = dict(launches=n, fetch_bytes_per_launch=int(fb), write_bytes_per_launch=int(wb))
        if name in dur:
            e["avg_us"] = round(dur[name], 2)
            e["hbm_gbps"] = round((fb + wb) / dur[name] / 1e3, 1)
        out["kernels"][name] = e
    json.dump(out, open(os.path.join(PR, f"r{RND}_{cfg}_pmc_traffic.json"), "w"), indent=1)
    sq = {}
    for sub in ("sqA", "sqB", "sqC"):
        for path in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
            for (name, cn), (n, v) in counters(path).items():
                sq.setdefault(name, {"launches": n})[cn] = round(v / n, 1)
    if sq:
        json.dump({"note": f"per-launch averages of SQ / GRBM counters, bench.py --config {cfg}, three rocprofv3 --pmc passes "
                           "(tools/evidence_round.sh); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over "
                           "waves, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md)", "kernels": sq},
                  open(os.path.join(PR, f"r{RND}_{cfg}_sq_counters.json"), "w"), indent=1)
    top = sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1]["launches"])[:6]
    print(cfg, json.load(open(os.path.join(d, "bench.json")))["value"], "it/s")
    for name, e in top:
        print(f"   {name[:48]:48s} {e.get('avg_us', 0):9.1f} us  fetch {e['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {e['write_bytes_per_launch'] / 1e6:9.1f} MB  {e.get('hbm_gbps', 0):7.1f} GB/s")
