"""CPU study (NumPy): tools/pf2_rounding_study.py on the iterates where tests/test_gpu_end_to_end.py measures config 4's margin
(third outer iteration, L2 ball active: |U| / |B| = 1.7) and with the engine's ACTUAL arithmetic (r x r products exact, vectors and
storage fp32).  Question: does an exactly stored L2-ball dual U buy parity margin?  Answer: no - 2.61e-7 -> 2.60e-7
per B-phase; the fp32 right-hand-side vector is what remains (exact: 2.08e-7), then the fp32 storage of B and P (1.67e-7).
    python tools/pf2_rounding_iterates.py"""
import sys, copy
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import pf2_rounding_study as S
from oracle import aoadmm_oracle as orc
I=48
J = np.random.RandomState(0).randint(128, 1025, I)
regs = [[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []]
X, rp = orc.synthetic_problem(I, J, 256, 16, seed=0, dtype=np.float64)
st = orc.random_state_for(S.f32(X), rp, 16, regs, seed=1)
for name in ("A","B","C"): setattr(st,name,S.f32(getattr(st,name)))
st.aux[1][0] = (S.f32(st.aux[1][0][0]), S.f32(st.aux[1][0][1]))
st.dual[1][0], st.aux[1][1], st.dual[1][1] = S.f32(st.dual[1][0]), S.f32(st.aux[1][1]), S.f32(st.dual[1][1])
# advance two outer iterations in the reference so that the ball is active (as on the iterates of the test)
for _ in range(2):
    st.update_B(); st.update_C(); st.update_A()
    for name in ("A","B","C"): setattr(st,name,S.f32(getattr(st,name)))
    st.aux[1][0] = (S.f32(st.aux[1][0][0]), S.f32(st.aux[1][0][1]))
    st.dual[1][0], st.aux[1][1], st.dual[1][1] = S.f32(st.dual[1][0]), S.f32(st.aux[1][1]), S.f32(st.dual[1][1])
ref = copy.deepcopy(st); ref.update_B()
err = lambda a,b: np.linalg.norm(a-b)/np.linalg.norm(b)
sites = ["xc_acc","xc_store","linv","v","solve_acc","y_sum","t_store","p_acc","z_acc","z_store"]
eng = {k: False for k in ("linv","solve_acc","y_sum","t_store","p_acc","z_acc","xc_acc")}  # the R64 engine: products exact, vectors / storage fp32
def rep(label, flags):
    B,Z,U = S.b_phase(st, flags); print(f"{label:60s} B {err(B,ref.B):.2e}", flush=True)
print("|U_l2| / |B| =", np.linalg.norm(st.dual[1][1])/np.linalg.norm(st.B))
rep("engine-like (products exact; v, stores fp32)", eng)
rep("engine-like + U_l2 stored exactly", dict(eng, st_Ul2=False))
rep("engine-like + U_l2 exact + v exact", dict(eng, st_Ul2=False, v=False))
rep("engine-like + v exact", dict(eng, v=False))
rep("engine-like + U_l2, Z_l2 exact + v exact", dict(eng, st_Ul2=False, st_Zl2=False, v=False))
rep("all compute exact, storage fp32", {k: False for k in sites})
