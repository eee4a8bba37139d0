# HBM bytes of the unimodal kernel forms on steady-state inputs (GPU box): bash tools/uni_pmc.sh [config] [iterations]
cd /tmp; export TMPDIR=/tmp
R=/root/repo; C=${1:-c5}; N=${2:-30}; O=$R/gpurun_out/uni_pmc; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $R/tools/uni_ab.py $C $N > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $R/tools/uni_ab.py $C $N > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
python3 - $O <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
def rows(sub, counter):
    path = glob.glob(os.path.join(O, sub, "*counter_collection.csv"))[0]
    out = [(int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(path))
           if "k_slab_unimodal" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sorted(out)
f, w = rows("fetch", "FETCH_SIZE"), rows("write", "WRITE_SIZE")
# the stand-alone prox calls of tools/uni_ab.py are the launches after the outer iterations: the last 9 + 5 x 13 x 2 ... take
# the launches by form in dispatch order and print the LAST 3 of every consecutive run of the same kernel name
def runs(rs):
    out, cur = [], []
    for r in rs:
        if cur and cur[-1][1] != r[1]:
            out.append(cur); cur = []
        cur.append(r)
    if cur: out.append(cur)
    return out
for (rf, rw) in zip(runs(f), runs(w)):
    name = rf[0][1].split("(")[0][-28:]
    fb = [2 * 1024 * v for _, _, v in rf]; wb = [1024 * v for _, _, v in rw]
    print(f"{name:30s} launches {len(rf):4d}: fetched (x2) first {fb[0]/1e9:7.2f} GB  last3 {[round(v/1e9,2) for v in fb[-3:]]}  written first {wb[0]/1e9:7.2f} last3 {[round(v/1e9,2) for v in wb[-3:]]}")
PY
for d in fetch write; do rm -f $O/$d/*kernel_trace.csv $O/$d/*agent_info.csv; python3 - $O/$d <<'PY'
import csv, glob, os, sys
for path in glob.glob(os.path.join(sys.argv[1], "*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(path)) if "k_slab_unimodal" in r["Kernel_Name"]]
    with open(path, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
done
