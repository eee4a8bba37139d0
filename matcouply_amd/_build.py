"""Builds libmatcouply_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmatcouply_hip.so")
SOURCES = ["contract.hip", "admm.hip", "generic.hip", "unimodal.hip", "sweep.hip", "reconstruct.hip", "wide.hip", "svdinit.hip", "cond.hip", "rowchain.hip", "xclds.hip", "api.hip"]
# per-file compiler options.  The kernel files: let small MFMA results live in VGPRs - by default the register allocator parks
# the 4-register accumulators of e.g. the sweep's inner loop / X C product in AGPRs and copies them back for every VALU use
# (468 v_accvgpr_* instructions in the config-3 sweep, 129 with the option; no scratch either way)
EXTRA_FLAGS = {f: ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] for f in ("sweep.hip", "contract.hip", "generic.hip", "admm.hip", "unimodal.hip", "rowchain.hip", "xclds.hip")}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "matcouply_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True, defs=None, out_lib=None, build_dir=None, only=None):
    """Compile the sources and link the library.  `defs` (default: $MCL_BUILD_DEFS) are extra compiler options, e.g.
    -DMCL_NO_ENV_SWITCHES for a release build; `out_lib` / `build_dir` put the result somewhere else (tests); `only` lists the
    sources to compile into `build_dir`, the other objects are taken from the in-tree build (compiled first if missing)."""
    out_lib = out_lib or LIB
    if out_lib == LIB and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if defs is None:
        defs = os.environ.get("MCL_BUILD_DEFS", "").split()  # e.g. -DMCL_NS_STAMPS: instrumented builds of the tools/, never shipped
    tree_dir = os.path.join(HERE, "build")
    build_dir = build_dir or tree_dir
    objs = []
    procs = []
    os.makedirs(build_dir, exist_ok=True)
    # an object is current when it is newer than ITS source and every header (the headers define the context's layout: an
    # object compiled against an older one may have another idea of it and is never linked); the other sources do not matter
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "matcouply_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    for src in SOURCES:
        name = src.replace(".hip", ".o")
        tree_obj = os.path.join(tree_dir, name)
        newest_dep = max(newest_header, os.path.getmtime(os.path.join(CSRC, src)))
        current = os.path.exists(tree_obj) and os.path.getmtime(tree_obj) >= newest_dep
        if current and ((only is not None and src not in only) or (only is None and not force and not defs and build_dir == tree_dir)):
            objs.append(tree_obj)  # unchanged by `defs` and not older than its source / any header: the in-tree object
            continue
        obj = os.path.join(build_dir, name)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + list(defs) + EXTRA_FLAGS.get(src, []) + [
            "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {src}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("building libmatcouply_hip.so failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out_lib


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
