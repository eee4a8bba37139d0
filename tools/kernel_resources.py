"""Register / scratch / LDS usage of every kernel of libmatcouply_hip.so as the compiler reports it
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950).  Usage: python tools/kernel_resources.py [file.hip ...] [--json out]"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "matcouply_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def resources(src):
    sys.path.insert(0, os.path.join(HERE, ".."))
    from matcouply_amd._build import EXTRA_FLAGS  # the per-file options the library is built with

    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(os.path.basename(src), []) + [
        "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows]) if rows else []
    for r, n in zip(rows, names):
        r["kernel"] = re.sub(r"^void ", "", n).split("(")[0]
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    allrows = []
    for f in files:
        path = f if os.path.exists(f) else os.path.join(CSRC, f)
        for r in resources(path):
            r["file"] = os.path.basename(path)
            allrows.append(r)
            print(f"{r['file']:13s} {r['kernel'][:70]:70s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} "
                  f"spill {r.get('VGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} "
                  f"occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?')}")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(allrows, fh, indent=1)
