"""Bit-level fingerprint of the unimodal prox on the seeded inputs of tools/unimodal_bench.py (to compare two builds of the
library across gpurun calls): python tools/uni_checksum.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unimodal_bench as ub

for I, data in ((1024, "noise"), (1024, "peak"), (2304, "noise"), (2304, "peak")):
    eng, aux = ub.make(I, 2048, 32, data, torch.device("cuda", 0))
    eng.B_begin(); eng.B_factor()
    eng.B_prox_local(0)
    torch.cuda.synchronize()
    h = hashlib.sha256(aux.cpu().numpy().tobytes()).hexdigest()[:16]
    d = hashlib.sha256(eng.regs[1][0].dual.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"I={I} data={data}: aux {h} dual {d}", flush=True)
    eng.close()
