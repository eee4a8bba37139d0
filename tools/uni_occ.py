"""Is the pruned unimodal kernel bound by latency or by throughput?  The prox on frozen steady-state inputs of a config-5 stack
at different occupancies of the SAME code (unused dynamic LDS limits the workgroups per CU; needs a -DMCL_UNI_DBG build):
    MCL_TEST_LIB=build_ab/<lib>.so python tools/uni_occ.py <pads, comma separated> [config=c5] [iterations=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from matcouply_amd import _engine
if os.environ.get("MCL_TEST_LIB"):
    _engine.LIB_PATH = os.path.abspath(os.environ["MCL_TEST_LIB"])
import bench

pads = [int(v) for v in sys.argv[1].split(",")]
name = sys.argv[2] if len(sys.argv) > 2 else "c5"
n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cfg = bench.CONFIGS[name]
dev = torch.device("cuda", 0)
X, row_ptr, I_loc = bench.make_shard(cfg, 0, 1, dev)
eng = bench.make_engine(cfg, X, row_ptr, I_loc, 0, dev)
kuni = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "unimodal"][0]
reg = eng.regs[1][kuni]
for _ in range(n_it):
    eng.update_B(); eng.update_C_local(); eng.update_C_finish(); eng.update_A()
torch.cuda.synchronize()
B0, U0 = eng.B.clone(), reg.dual.clone()
eng.B_begin(); eng.B_factor()
os.environ["MCL_UNI_SPLIT"] = "0"
eng.reload_switches()
for pad in pads:
    os.environ["MCL_UNI_PAD_LDS"] = str(pad)
    ts = []
    for rep in range(3):
        eng.B.copy_(B0); reg.dual.copy_(U0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.B_prox_local(kuni); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"  {os.path.basename(_engine.LIB_PATH)} pad {pad:6d} B of LDS: " + " ".join(f"{t:8.3f}" for t in ts) + " ms", flush=True)
