import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import aoadmm_oracle as orc
from tests.helpers import engine_from_oracle_state, rel_err, to_np

J = np.array([300, 64, 1100, 257])
X, row_ptr = orc.synthetic_problem(4, J, 256, 16, seed=1, dtype=np.float64)
X = X.astype(np.float32).astype(np.float64)
nn = {"kind": "nn"}
st = orc.random_state_for(X, row_ptr, 16, [[nn], [nn], [nn]], seed=2)
import copy
ref = copy.deepcopy(st)
eng = engine_from_oracle_state(st)
print("variant before", eng.kernel_variant(3) if hasattr(eng, "kernel_variant") else "")
eng.update_B(); torch.cuda.synchronize()
ref.update_B()
print("B", rel_err(to_np(eng.B), ref.B), "aux", rel_err(to_np(eng.regs[1][0].aux), ref.aux[1][0]), "dual", rel_err(to_np(eng.regs[1][0].dual), ref.dual[1][0]))
d = np.abs(to_np(eng.B) - ref.B).max(axis=1)
bad = np.nonzero(d > 1e-4 * np.abs(ref.B).max())[0]
print("bad rows", len(bad), bad[:40])
gr = eng.update_C_local(); torch.cuda.synchronize()
r = 16
Ba = np.concatenate([ref.B[row_ptr[i]:row_ptr[i+1]] * ref.A[i] for i in range(4)])
G = Ba.T @ Ba; R = X.T @ Ba
g = to_np(gr)
print("G", rel_err(g[:r*r].reshape(r, r), G), "R", rel_err(g[r*r:].reshape(-1, r), R))
eng.update_C_finish(); ref.update_C(); torch.cuda.synchronize()
print("C", rel_err(to_np(eng.C), ref.C))
eng.update_A(); ref.update_A(); torch.cuda.synchronize()
print("A", rel_err(to_np(eng.A), ref.A), "rhses", rel_err(to_np(eng.rhses()), ref.rhses if hasattr(ref, 'rhses') else to_np(eng.rhses())))
print("variant", eng.kernel_variant(3) if hasattr(eng, "kernel_variant") else "")
