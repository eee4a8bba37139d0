"""Same-box A/B of the unimodal prox between two BUILDS of the library on the steady-state iterates of a config-5 stack:
    python tools/uni_lib_ab.py <libA.so> <libB.so> [config=c5] [iterations=30]
Each library runs in its own process (`--one`): the stack is iterated to the given outer iteration with THAT library (the builds
are expected to agree to the bit there: the checksum of the frozen input is printed), then the prox alone is timed 5 times and
the result hashed; finally the whole-iteration rate."""
import hashlib, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] != "--one":
    libs = sys.argv[1:3]
    rest = sys.argv[3:]
    for lib in libs:
        subprocess.run([sys.executable, __file__, "--one", lib] + rest, check=True)
    sys.exit(0)

lib = sys.argv[2]
name = sys.argv[3] if len(sys.argv) > 3 else "c5"
n_it = int(sys.argv[4]) if len(sys.argv) > 4 else 30
import numpy as np, torch
from matcouply_amd import _engine
_engine.LIB_PATH = os.path.abspath(lib)
import bench

cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
kuni = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "unimodal"][0]
reg = eng.regs[1][kuni]


def iterate(n):
    for _ in range(n):
        eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()


iterate(n_it)
torch.cuda.synchronize()
B0, U0 = eng.B.clone(), reg.dual.clone()
h_in = hashlib.sha256((B0 + U0).cpu().numpy().tobytes()).hexdigest()[:16]
eng.B_begin(); eng.B_factor()
os.environ["MCL_UNI_SPLIT"] = "0"
eng.reload_switches()
ts = []
for rep in range(5):
    eng.B.copy_(B0); reg.dual.copy_(U0); reg.aux.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.B_prox_local(kuni); e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
h = hashlib.sha256(reg.aux.cpu().numpy().tobytes()).hexdigest()[:16]
eng.B.copy_(B0); reg.dual.copy_(U0)
eng.B_end()
os.environ.pop("MCL_UNI_SPLIT")
eng.reload_switches()
torch.cuda.synchronize()
t0 = time.perf_counter(); iterate(10); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{os.path.basename(lib):36s} input {h_in}  prox " + " ".join(f"{t:7.3f}" for t in ts) + f" ms  aux {h}   whole iteration {dt * 1e3:8.3f} ms", flush=True)
