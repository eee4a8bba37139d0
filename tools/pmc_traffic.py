"""rocprofv3 counter CSVs (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes) -> profiles/r1_<cfg>_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<note>"
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports 1/2 of a wide coalesced streaming read; unit KiB); WRITE_SIZE
is exact (unit KiB)."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"note": sys.argv[4], "kernels": {}}
for name, (n, v) in fetch.items():
    if name.startswith("k_") or name.startswith("void k_"):
        w = write.get(name, [n, 0.0])
        out["kernels"][name] = dict(launches=n, fetch_bytes_per_launch=int(2 * 1024 * v / n),
                                    write_bytes_per_launch=int(1024 * w[1] / max(w[0], 1)))
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1)[:1500])
