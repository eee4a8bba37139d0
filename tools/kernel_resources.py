"""Register / scratch / LDS usage of every kernel of libmatcouply_hip.so, read from the code objects INSIDE the built library
(the gfx950 ELFs of its .hip_fatbin section; their amdhsa.kernels metadata via llvm-readelf) - what ships, in a second or two.

    python tools/kernel_resources.py                      # table of all kernels
    python tools/kernel_resources.py --json out.json      # the same as JSON
    python tools/kernel_resources.py --update             # rewrite profiles/kernel_resources.json (the committed baseline
                                                          #   tests/test_kernel_resources.py holds the hot kernels to)
Occupancy is the register-limited one: min(8, 512 // roundup(vgpr_count, 8)) waves per SIMD (gfx950: 512 unified VGPRs per
lane and SIMD, `.vgpr_count` = architectural + accumulation registers); LDS limits are listed as bytes per workgroup."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB = os.path.join(REPO, "matcouply_amd", "libmatcouply_hip.so")
BASELINE = os.path.join(REPO, "profiles", "kernel_resources.json")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
# the kernels the iteration time of the BASELINE configurations is made of: held to the committed baseline
HOT = [r"^k_sweep<", r"^k_contract_x", r"^k_rows_finish_solve_stats<", r"^k_rows_finish_fused<", r"^k_rows_solve_stats<",
       r"^k_pf2_algebra_ns<", r"^k_slab_unimodal_v4<", r"^k_rows_fused<", r"^k_reduce_frag", r"^k_A_finish_rows", r"^k_C_finish_multi",
       r"k_rows_chain_(first|mid|last)<"]


def code_objects(lib=LIB):
    """the gfx950 ELF images bundled in the library (one per translation unit)"""
    data = open(lib, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24: p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off: pos + off + size])
        pos += len(MAGIC)
    return out


def demangle(names):
    if not names:
        return []
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def resources(lib=LIB):
    rows = []
    for image in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            block = ".agpr_count:" + block
            get = lambda key, d=0: (lambda m: int(m.group(1)) if m else d)(re.search(r"\.%s:\s+(\d+)" % key, block))
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name:
                continue
            rows.append(dict(mangled=name.group(1), vgpr_count=get("vgpr_count"), agpr_count=get("agpr_count"),
                             sgpr_count=get("sgpr_count"), vgpr_spill=get("vgpr_spill_count"), sgpr_spill=get("sgpr_spill_count"),
                             scratch_bytes=get("private_segment_fixed_size"), lds_bytes=get("group_segment_fixed_size"),
                             max_workgroup=get("max_flat_workgroup_size")))
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["kernel"] = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").split("(")[0]
        r["occupancy"] = min(8, 512 // max(8, -(-r["vgpr_count"] // 8) * 8))
    return sorted(rows, key=lambda r: r["kernel"])


def is_hot(kernel):
    return any(re.search(p, kernel) for p in HOT)


if __name__ == "__main__":
    rows = resources()
    if "--update" in sys.argv:
        keep = {r["kernel"]: {k: r[k] for k in ("vgpr_count", "agpr_count", "vgpr_spill", "scratch_bytes", "lds_bytes", "occupancy")}
                for r in rows}
        json.dump({"note": "tools/kernel_resources.py --update: amdhsa.kernels metadata of the gfx950 code objects in "
                           "libmatcouply_hip.so (hipcc, ROCm 7.2); `hot` kernels are held to these numbers by "
                           "tests/test_kernel_resources.py", "hot": sorted(k for k in keep if is_hot(k)), "kernels": keep},
                  open(BASELINE, "w"), indent=1, sort_keys=True)
        print("wrote", BASELINE, len(keep), "kernels,", sum(is_hot(k) for k in keep), "hot")
    elif "--json" in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    else:
        for r in rows:
            print(f"{r['kernel'][:84]:84s} vgpr {r['vgpr_count']:4d} (agpr {r['agpr_count']:3d}) spill {r['vgpr_spill']:3d} "
                  f"scratch {r['scratch_bytes']:5d} lds {r['lds_bytes']:6d} occ {r['occupancy']}{'  *' if is_hot(r['kernel']) else ''}")
