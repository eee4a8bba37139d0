"""A/B of the unimodal-regression kernel forms on the STEADY-STATE iterates of a config-5 stack (GPU box):
    python tools/uni_ab.py [config=c5] [iterations=30]
Runs the stack to the given outer iteration, freezes B and the dual of the unimodal penalty, then times the prox alone in each
form on those inputs (HIP events, 3 repetitions) and compares the fits bit by bit; finally the whole-iteration rate per form."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
print(f"{name}: {n_it} outer iterations done; {X.shape[0] * cfg['r'] / 1e6:.1f} M elements per call", flush=True)
B0, U0 = eng.B.clone(), reg.dual.clone()
eng.B_begin(); eng.B_factor()
forms = [("no pruning", {"MCL_UNI_SPLIT": "0", "MCL_UNI_NOPRUNE": "1"}),
         ("pruned sweeps (default)", {"MCL_UNI_SPLIT": "0"})]
if len(sys.argv) > 3 and sys.argv[3] == "latency":
    forms.append(("latency form (MODE 1 + 2)", {"MCL_UNI_SPLIT": "1"}))
ref = None
for label, env in forms:
    for k in ("MCL_UNI_SPLIT", "MCL_UNI_NOPRUNE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng.reload_switches()
    ts = []
    for rep in range(3):
        eng.B.copy_(B0); reg.dual.copy_(U0); reg.aux.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.B_prox_local(kuni); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    h = hashlib.sha256(reg.aux.cpu().numpy().tobytes()).hexdigest()[:16]
    if ref is None:
        ref = reg.aux.clone()
    nd = int((ref != reg.aux).sum())
    print(f"  {label:48s} " + " ".join(f"{t:8.3f}" for t in ts) + f" ms   aux {h}  differs from the first form in {nd} elements"
          + (f" (max {float((ref - reg.aux).abs().max()):.2e})" if nd else ""), flush=True)
eng.B.copy_(B0); reg.dual.copy_(U0)
eng.B_end()
for label, env in (("no pruning", {"MCL_UNI_NOPRUNE": "1"}), ("default", {})):
    for k in ("MCL_UNI_SPLIT", "MCL_UNI_NOPRUNE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng.reload_switches()
    iterate(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    iterate(10)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"  whole iteration, {label:18s}: {1e3 * dt:8.3f} ms = {1 / dt:7.2f} it/s", flush=True)
