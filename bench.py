#!/usr/bin/env python3
"""bench.py - AO-ADMM outer iterations/s of the MI355X engine on BASELINE.json's metric configuration.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE outer AO-ADMM iteration (B-phase -> C-phase -> A-phase + per-iteration diagnostics, i.e.
`return_errors=True` as in the reference's default loop) of `cmf_aoadmm(non_negative=True, l1_penalty={2: 0.1})`
on synthetic data of config 3 (I=1024, J_i=512, K=256, rank 16), inputs resident in HBM.
With N > 1 the I slabs are sharded over the ranks in contiguous ranges balanced by their ROWS (partition_slabs; fixed
total problem -> "strong" scaling, as BASELINE.json's metric "@1/2/4/8 GPU" states); per step there is one RCCL
all-reduce of the fp64 C-mode normal equations [G | R] (PARAFAC2 stacks add one small all-reduce per inner iteration);
the fp64 diagnostic sums of all steps are all-reduced once at the end of each timed region (no stopping rule is active
in this fixed-iteration workload).  After the run the replicated factor C is checked to be BIT-identical on all ranks.

Timing: W warm-up steps, then R (default 5) timed regions of exactly K steps each, every region bracketed by a barrier
+ torch.cuda.synchronize(); per region the MAX over ranks; the reported `ms_per_step * steps` is the MEDIAN region
(`region_ms` lists all of them) - one region of 20 steps lasts ~4 ms and a single sample of that swings by 10 %.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      dominant kernel's algorithmic bytes / HIP-event time (events recorded inside the library on the
                kernel's own stream) against the 8 TB/s HBM peak
  cpu_baseline  the NumPy oracle (oracle/aoadmm_oracle.py, a restatement of the reference pinned by goldens)
                timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # name: (I, J, K, r, regs per mode as descriptor lists)
    "c2": dict(I=256, J=256, K=128, r=8, regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "nn"}]],
               desc="c2: I=256 J_i=256 K=128 rank=8, non_negative on all modes"),
    "c3": dict(I=1024, J=512, K=256, r=16,
               regs=[[{"kind": "nn"}], [{"kind": "nn"}], [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]],
               desc="c3: I=1024 J_i=512 K=256 rank=16, non_negative + L1(0.1) on C"),
}
CONFIGS["c4"] = dict(I=1024, J="ragged", K=256, r=16,
                     regs=[[], [{"kind": "parafac2"}, {"kind": "l2ball", "norm_bound": 1.0}], []],
                     desc="c4: I=1024 ragged J_i in [128,1024] K=256 rank=16, parafac2 + L2Ball(1.0) on B")
CONFIGS["c5s"] = dict(I=1024, J=512, K=256, r=32,
                      regs=[[{"kind": "nn"}],
                            [{"kind": "parafac2"}, {"kind": "unimodal", "non_negativity": True},
                             {"kind": "l2ball", "norm_bound": 1.0, "non_negativity": True}],
                            [{"kind": "l1", "reg_strength": 0.1, "non_negativity": True}]],
                      desc="c5 penalty stack (NN + L1 + L2Ball + Unimodal + PARAFAC2) at I=1024 J_i=512 K=256 rank=32 "
                           "(config 5 itself is I=8192 J=2048 K=1024: 68.7 GB of X)")
CONFIGS["c5"] = dict(CONFIGS["c5s"], I=8192, J=2048, K=1024, desc="c5: I=8192 J_i=2048 K=1024 rank=32 fp32 (X = 68.7 GB), "
                     "NN + L1 + L2Ball + Unimodal + PARAFAC2; fits one MI355X (128 GB in use), shards over --gpus N")
CONFIGS["c5_8th"] = dict(CONFIGS["c5"], I=1024, desc="one eighth of config 5 (the per-rank shard of an 8-GPU run): I=1024 J_i=2048 K=1024 rank=32")
CONFIGS["c5_32nd"] = dict(CONFIGS["c5s"], I=256, J=2048, K=1024, desc="1/32 of config 5: I=256 J_i=2048 K=1024 rank=32, full penalty stack")
CONFIGS["c3_8th"] = dict(CONFIGS["c3"], I=128, desc="one eighth of config 3 (the per-rank shard of an 8-GPU run): I=128 J_i=512 K=256 rank=16")
CONFIGS["c3_half"] = dict(CONFIGS["c3"], I=512, desc="one half of config 3 (the per-rank shard of a 2-GPU run): I=512 J_i=512 K=256 rank=16")
CONFIGS["c3_4th"] = dict(CONFIGS["c3"], I=256, desc="one quarter of config 3 (the per-rank shard of a 4-GPU run): I=256 J_i=512 K=256 rank=16")
CONFIGS["c3_64th"] = dict(CONFIGS["c3"], I=16, J=256, desc="a small problem (1 M elements of X: the exact-products mode of the library): I=16 J_i=256 K=256 rank=16")
CONFIGS["c3r"] = dict(CONFIGS["c3"], J="ragged", desc="config 3 with the ragged slabs of config 4: I=1024 J_i in [128,1024] K=256 rank=16")
CONFIGS["k512"] = dict(CONFIGS["c3"], I=512, J=512, K=512, desc="K=512 variant of config 3 (same bytes of X): I=512 J_i=512 K=512 rank=16")
CONFIGS["r32"] = dict(CONFIGS["c3"], r=32, desc="rank-32 variant of config 3: I=1024 J_i=512 K=256 rank=32")
# the same penalty stacks as keyword arguments of the public API (the `api` block of the JSON line)
CONFIGS["c2"]["api_kwargs"] = dict(non_negative=True)
CONFIGS["c3"]["api_kwargs"] = dict(non_negative=True, l1_penalty={2: 0.1})
CONFIGS["c4"]["api_kwargs"] = dict(parafac2=True, l2_norm_bound={1: 1.0})
_T0 = time.perf_counter()
WALL_S = {}  # wall time of every leg of this command (rank 0), printed on stderr as it goes and in the JSON line


def leg(name, t_start):
    """Record the wall time of one leg and say so on stderr (the JSON line on stdout stays the only stdout output)."""
    dt = time.perf_counter() - t_start
    WALL_S[name] = round(WALL_S.get(name, 0.0) + dt, 3)
    if int(os.environ.get("RANK", "0")) == 0:
        try:  # bytes this process has read through read() so far (files, libraries loaded with read): names a cold-cache leg
            with open("/proc/self/io") as f:
                rchar = int(f.read().split("rchar:")[1].split()[0]) / 1e6
        except (OSError, ValueError, IndexError):
            rchar = float("nan")
        print(f"[bench] {name}: {dt:.2f} s (t+{time.perf_counter() - _T0:.1f} s, {rchar:.0f} MB read so far)", file=sys.stderr, flush=True)
    return time.perf_counter()


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix rate (v_mfma_f32_16x16x4_f32: 256 flop/cycle/CU)


def make_shard(cfg, rank, world, device, seed=0):
    """Synthetic X_i = B_i* diag(a_i*) C*^T + 0.05 N(0,1) for this rank's contiguous slab range."""
    import torch

    I, J, K, r = cfg["I"], cfg["J"], cfg["K"], cfg["r"]
    from matcouply_amd._engine import cmf_to_packed
    from matcouply_amd.decomposition import partition_slabs

    J_all = np.random.RandomState(0).randint(128, 1025, I) if J == "ragged" else np.full(I, J)
    mine = partition_slabs(J_all, world)[rank]  # contiguous range, balanced by rows (sum of J_i), not by slab count
    lo, hi = (int(mine[0]), int(mine[-1]) + 1) if len(mine) else (0, 0)
    I_loc = hi - lo
    if J == "ragged":
        return make_ragged_shard(cfg, lo, hi, rank, device, seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    C_true = torch.rand((K, r), generator=g, device=device)
    g.manual_seed(seed + 1000 + rank)
    row_ptr = np.arange(I_loc + 1, dtype=np.int64) * J
    if 4.0 * I_loc * J * K > 8e9:  # config 5 (68.7 GB of X): generated 64 slabs at a time, unimodal non-negative B_i*
        X = torch.empty((I_loc * J, K), dtype=torch.float32, device=device)
        t = torch.linspace(0, 1, J, device=device)[None, :, None]
        for i0 in range(0, I_loc, 64):
            n = min(64, I_loc - i0)
            A_t = torch.rand((n, r), generator=g, device=device) + 0.1
            mu = torch.rand((n, 1, r), generator=g, device=device)
            sig = 0.05 + 0.2 * torch.rand((n, 1, r), generator=g, device=device)
            B_t = torch.exp(-0.5 * ((t - mu) / sig) ** 2).reshape(n * J, r).contiguous()
            Xc = cmf_to_packed(A_t, B_t, C_true, row_ptr[:n + 1])
            Xc += 0.05 * torch.randn(Xc.shape, generator=g, device=device)
            X[i0 * J:(i0 + n) * J] = Xc
            del Xc
        return X, row_ptr, I_loc
    A_true = torch.rand((I_loc, r), generator=g, device=device) + 0.1
    B_true = torch.rand((I_loc * J, r), generator=g, device=device)
    # the library's own dense-reconstruction kernel (mcl_cmf_to_packed), not a torch GEMM: the bench then never loads
    # rocBLAS / hipBLASLt and their kernel libraries (hundreds of MB to page in on a cold box)
    X = cmf_to_packed(A_true, B_true, C_true, row_ptr)
    X += 0.05 * torch.randn(X.shape, generator=g, device=device)
    return X, row_ptr, I_loc


def make_ragged_shard(cfg, lo, hi, rank, device, seed):
    """config 4: J_i = RandomState(0).randint(128, 1025, I)"""
    import torch

    K, r = cfg["K"], cfg["r"]
    J_all = np.random.RandomState(0).randint(128, 1025, cfg["I"])
    J = J_all[lo:hi]
    row_ptr = np.concatenate([[0], np.cumsum(J)]).astype(np.int64)
    N = int(row_ptr[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    C_true = torch.rand((K, r), generator=g, device=device)
    g.manual_seed(seed + 1000 + rank)
    A_true = torch.rand((hi - lo, r), generator=g, device=device) + 0.1
    B_true = torch.rand((N, r), generator=g, device=device)
    from matcouply_amd._engine import cmf_to_packed

    X = cmf_to_packed(A_true, B_true, C_true, row_ptr)
    X += 0.05 * torch.randn(X.shape, generator=g, device=device)
    return X, row_ptr, hi - lo


def make_engine(cfg, X, row_ptr, I_loc, rank, device, seed=1):
    import torch
    from matcouply_amd._engine import KIND, HipEngine, NativeReg

    K, r, N = cfg["K"], cfg["r"], X.shape[0]
    g = torch.Generator(device=device)
    g.manual_seed(seed)  # replicated quantities (C and its ADMM variables) are drawn identically on all ranks
    C = torch.rand((K, r), generator=g, device=device)
    shapes = {0: (I_loc, r), 1: (N, r), 2: (K, r)}
    c_vars = [(torch.rand((K, r), generator=g, device=device), torch.rand((K, r), generator=g, device=device))
              for _ in cfg["regs"][2]]
    g.manual_seed(seed + 1000 + rank)
    A = torch.rand((I_loc, r), generator=g, device=device)
    B = torch.rand((N, r), generator=g, device=device)
    regs = [[], [], []]
    for m in range(3):
        for k, d in enumerate(cfg["regs"][m]):
            aux2 = None
            if m == 2:
                aux, dual = c_vars[k]
            elif d["kind"] == "parafac2":
                aux = torch.zeros(shapes[m], device=device)  # P_i = eye(J_i, r)
                rp = torch.as_tensor(row_ptr[:-1], device=device)
                for cc in range(r):
                    aux[rp + cc, cc] = 1.0
                dual = torch.rand(shapes[m], generator=g, device=device)
                g2 = torch.Generator(device=device)
                g2.manual_seed(seed + 7)  # Delta is replicated: identical on all ranks
                aux2 = torch.rand((r, r), generator=g2, device=device)
            else:
                aux = torch.rand(shapes[m], generator=g, device=device)
                dual = torch.rand(shapes[m], generator=g, device=device)
            regs[m].append(NativeReg(KIND[d["kind"]], aux, dual, aux2=aux2, non_negativity=d.get("non_negativity", False),
                                     p0=d.get("reg_strength", d.get("norm_bound", 0.0))))
    return HipEngine(X, row_ptr, r, A, B, C, regs)


def usable_cores():
    """Host cores this process may use: the affinity mask capped by the cgroup CPU quota (cpu.max), if any."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(cfg, budget_s=15.0):
    """Oracle (NumPy, fp64 like the reference) on the host cores, bounded sample: a subset of the slabs, >= 1 iteration."""
    t_leg = time.perf_counter()
    from oracle import aoadmm_oracle as orc

    cores = usable_cores()
    try:  # BLAS threads = the cores this process may actually use (a 256-thread pool on a 16-CPU quota thrashes)
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=cores)
    except Exception:
        pass
    I, J, K, r = cfg["I"], cfg["J"], cfg["K"], cfg["r"]

    J_all = np.random.RandomState(0).randint(128, 1025, I) if J == "ragged" else None  # config 4

    def timed(I_s, iters, tag):
        t_sub = time.perf_counter()
        X, row_ptr = orc.synthetic_problem(I_s, J if J_all is None else J_all[:I_s], K, r, seed=0, dtype=np.float64)
        st = orc.random_state_for(X, row_ptr, r, cfg["regs"], seed=1)
        t_sub = leg(f"cpu_baseline.{tag}.synthesis", t_sub)
        st.update_B(); st.update_C(); st.update_A()  # warm-up iteration (BLAS thread pool, page faults)
        t_sub = leg(f"cpu_baseline.{tag}.warmup_iteration", t_sub)
        t0 = time.perf_counter()
        for _ in range(iters):
            st.update_B(); st.update_C(); st.update_A()
            st.feasibility_gaps(); st.loss(st.rec_error_from_A_byproducts())
        leg(f"cpu_baseline.{tag}.timed_iterations", t_sub)
        return (time.perf_counter() - t0) / iters

    t_leg = leg("cpu_baseline.import+threadpool", t_leg)
    I_probe = min(I, 64)
    t_probe = timed(I_probe, 1, "probe")
    per_slab = t_probe / I_probe
    I_s = int(min(I, max(I_probe, budget_s / 3.0 / max(per_slab, 1e-9))))
    # about two thirds of the budget in timed iterations (the brief: a bounded sample of roughly 10 s or more of CPU work): 2 of
    # them when one iteration of the sample already takes a third of the budget (configs 4 / 5), up to 20 on the small ones
    iters = int(max(2, min(20, (budget_s * 0.67) / max(per_slab * I_s, 1e-9))))
    t_iter = timed(I_s, iters, "sample")
    value = 1.0 / (t_iter * I / I_s)
    survey = {"c3": 0.41, "c2": 7.5}.get(cfg.get("name"))
    return dict(value=value, unit="outer-iters/s", cores=cores, kind="port",
                note="the port is the vectorised NumPy restatement of the reference (oracle/, pinned by the reference's "
                     "goldens); the reference proper is slower (per-slab Python loops)",
                reference_survey=(dict(value=survey, unit="outer-iters/s", cores=8,
                                       where="unmodified reference, survey container (BASELINE.md section 2)")
                                  if survey else None),
                sample=f"{iters} outer iterations (after 1 warm-up) of the NumPy fp64 oracle on the first {I_s} of {I} slabs "
                       f"of the same synthetic workload, BLAS threads = {cores} (affinity mask capped by the cgroup CPU quota; "
                       f"os.cpu_count() = {os.cpu_count()}); value scaled by {I_s}/{I}")


def spawn_ranks(n):
    """Start `n` ranks of this script under torch.distributed.run as child processes; returns their exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:  # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr passes through
    lines = []
    for line in proc.stdout:
        lines.append(line)
    rc = proc.wait()
    json_lines = [l for l in lines if l.startswith("{")]
    for l in lines:
        if not l.startswith("{"):
            sys.stderr.write(l)  # anything else the ranks printed is not part of the contract line
    if json_lines:
        sys.stdout.write(json_lines[-1])
        sys.stdout.flush()
    return rc if rc else (0 if json_lines else 1)


def api_block(cfg, X, row_ptr, n_short=100, n_long=1000):
    """Outer iterations / s of the PUBLIC call - `cmf_aoadmm(PackedMatrices, rank, ...)` - on the resident data, with
    `tol=None` (fixed iteration count: mcl_iterate) and with the DEFAULT tolerances (tol=1e-8, absolute_tol=1e-10,
    feasibility_tol=1e-4: the stopping rule evaluated on the device, mcl_run).  Every call pays the set-up of the
    reference's API (host RNG of the initial factors and ADMM variables, upload, context): the rate is therefore taken
    from the DIFFERENCE of a long and a short call, (n_long - n_short) / (t_long - t_short)."""
    import torch
    from matcouply_amd import decomposition as dec

    kw = cfg.get("api_kwargs")
    if kw is None:
        return None
    packed = dec.PackedMatrices(X, row_ptr)

    def call(n, **tols):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, diag = dec.cmf_aoadmm(packed, cfg["r"], n_iter_max=n, random_state=0, return_errors=True, **kw, **tols)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"[bench] api call n_iter_max={n} {tols or 'default tolerances'}: {dt:.3f} s, n_iter {diag.n_iter}", file=sys.stderr, flush=True)
        return dt, diag.n_iter

    out = {"call": "cmf_aoadmm(PackedMatrices, rank=%d, %s, return_errors=True, random_state=0)" % (
        cfg["r"], ", ".join(f"{k}={v}" for k, v in kw.items())), "n_iter_max": [n_short, n_long]}
    call(3, tol=None, absolute_tol=None)  # first call: library / allocator warm-up
    for name, tols in (("tol_none", dict(tol=None, absolute_tol=None)), ("default_tol", dict())):
        # the set-up part of a call (host RNG, upload) varies by milliseconds from call to call and only ever adds time:
        # each length is timed `reps` times and the fastest call is kept
        reps = 2
        (t1, n1) = min(call(n_short, **tols) for _ in range(reps))
        (t2, n2) = min(call(n_long, **tols) for _ in range(reps))
        rate = (n2 - n1) / (t2 - t1) if n2 > n1 and t2 > t1 else n2 / t2
        out[name] = {"iters_per_s": round(rate, 1), "seconds": [round(t1, 4), round(t2, 4)], "n_iter": [n1, n2],
                     "from": "difference of the two calls (fastest of %d each)" % reps if n2 > n1 else "whole call (stopped early)"}
    out["default_over_tol_none"] = round(out["default_tol"]["iters_per_s"] / out["tol_none"]["iters_per_s"], 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--regions", type=int, default=25, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="un-timed settling after --warmup: keep stepping for at least this long AND until two consecutive "
                         "probe regions of --steps steps agree within 2 %% (clock ramp of a fresh device); 0 disables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api", action="store_true", help="skip the `api` block (timing of the public cmf_aoadmm call)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--wall-budget", type=float, default=240.0,
                    help="seconds this command should stay under: the two optional legs after the timed regions (the `api` "
                         "block, the CPU baseline) are shortened or skipped - and say so in the JSON line - when the mandatory "
                         "legs (imports, device, data, timed regions) have used the budget up, e.g. on a box with cold caches")
    args = ap.parse_args()

    # RCCL shares device buffers between the ranks of a node through dmabuf IPC; the legacy IPC mode is not supported by the
    # host driver of this pool (hipIpcGetMemHandle: invalid argument) - keep the setting the image exports
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU
        # (no torch import yet) and never will: the N ranks are CHILD processes of torch.distributed.run, rank 0's JSON
        # line is forwarded and the children's exit code is ours.  (No exec: replacing a process is not allowed on the
        # GPU boxes once anything has initialised the device, and a child keeps the parent free to report failures.)
        return spawn_ranks(args.gpus)
    t_leg = time.perf_counter()
    import torch
    import torch.distributed as dist

    t_leg = leg("import_torch", t_leg)

    cfg = dict(CONFIGS[args.config], name=args.config)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree "
                         f"(one rank per GPU; without a launcher `python bench.py --gpus N` starts its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # MCL_BENCH_SHARE_GPU=1 (+ MCL_BENCH_BACKEND=gloo): all ranks use device 0 - a functional check of the sharded path
    # on a single-GPU box (RCCL itself refuses two ranks on one device); never a performance configuration.
    if os.environ.get("MCL_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("MCL_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            # one node by contract (rendezvous on 127.0.0.1): RCCL's out-of-band bootstrap may use the loopback interface
            # too - a container without a routable interface (or with an unresolvable hostname) then cannot stall it
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    torch.zeros(1, device=device)
    torch.cuda.synchronize()
    t_leg = leg("device_init" + ("+process_group" if world > 1 else ""), t_leg)
    X, row_ptr, I_loc = make_shard(cfg, rank, world, device)
    torch.cuda.synchronize()
    t_leg = leg("data_synthesis", t_leg)
    eng = make_engine(cfg, X, row_ptr, I_loc, rank, device)
    torch.cuda.synchronize()
    t_leg = leg("engine_setup", t_leg)
    from matcouply_amd._engine import DIAG_LEN

    n_regions = max(1, args.regions)
    ring = torch.zeros((max(1, n_regions * args.steps), DIAG_LEN), dtype=torch.float64, device=device)
    scratch = torch.zeros((max(1, args.warmup, args.steps), DIAG_LEN), dtype=torch.float64, device=device)  # un-timed steps
    n_coll = [0]  # collectives issued by this rank (counted on the host)

    # collectives of the timed loop: RCCL called directly on the engine's stream (matcouply_amd/_rccl.py; self-tested
    # against torch.distributed at start-up, which stays the fallback and the path of every other backend)
    direct = None
    if world > 1 and backend == "nccl":
        from matcouply_amd._rccl import DirectComm

        direct = DirectComm.try_create(dist.group.WORLD)
        t_leg = leg("rccl_direct_comm", t_leg)

    def all_reduce(t, **kw):
        n_coll[0] += 1
        if direct is not None and not kw:
            direct.all_reduce(t)
        else:
            dist.all_reduce(t, **kw)

    pf2 = [k for k, d in enumerate(cfg["regs"][1]) if d["kind"] == "parafac2"]

    # MCL_BENCH_FORCE_STEPS=1: drive the B-phase through the step calls on ONE rank too (the path a multi-GPU run takes
    # when PARAFAC2 needs its per-inner-iteration all-reduce) - for measuring a rank's work on a single-GPU box
    force_steps = os.environ.get("MCL_BENCH_FORCE_STEPS") == "1"

    def update_B():
        if (world == 1 and not force_steps) or not pf2:
            eng.update_B()
            return
        eng.B_begin()
        eng.B_factor()
        for _ in range(5):
            eng.B_solve()
            for k in range(len(cfg["regs"][1])):
                eng.B_prox_local(k)
                if k in pf2 and world > 1:
                    all_reduce(eng.B_prox_reduce_buffer(k))
                eng.B_prox_finish(k)
        eng.B_end()

    def step(slot):
        update_B()
        gr = eng.update_C_local()
        if world > 1:
            all_reduce(gr)
        eng.update_C_finish()
        # no stopping rule is active (tol=None), so the per-iteration diagnostic sums stay on the device and are
        # all-reduced ONCE for all iterations at the end of the timed region (one collective per step remains: [G | R])
        eng.update_A()
        # deferred: the reduction of the diagnostics tables rides on the next step's C-phase reduction kernel (what
        # mcl_iterate - the fixed-count loop behind cmf_aoadmm - does between its iterations); flushed at the region end
        eng.diagnostics_deferred(include_replicated=(rank == 0), out=slot)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step(scratch[it])
    eng.flush_diagnostics()
    sync()
    t_leg = leg("warmup", t_leg)
    # Settling (un-timed, on top of --warmup): a fresh device ramps its clocks over the first ~100 ms of work - with 5
    # warm-up steps (0.8 ms at config 3) the first timed regions of round 2 fell monotonically by 10 %.  Keep stepping in
    # probe regions of --steps steps until >= --settle-ms have passed AND two consecutive probes agree within 2 %
    # (bounded by 3 s).  Every rank takes the same decision (the probe time is the MAX over ranks).
    settle = dict(ms=0.0, probes=0, settled=None)
    if args.settle_ms > 0 and args.steps > 0:
        prev, total = None, 0.0
        for probe in range(1000000):
            sync()
            t0 = time.perf_counter()
            for it in range(args.steps):
                step(scratch[it])
            eng.flush_diagnostics()
            sync()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            total += dt
            agree = prev is not None and abs(dt - prev) <= 0.02 * max(dt, prev)
            prev = dt
            settle = dict(ms=round(1e3 * total, 2), probes=probe + 1, settled=bool(agree))
            if (1e3 * total >= args.settle_ms and agree) or total > 3.0:
                break
    # HIP events inside the library around every `stride`-th launch of the timed region: an event pair opens ~5 us
    # dispatch gaps before and after the kernel (11 us per step when every launch is bracketed - measured), so the
    # timed loop samples ~10 launches instead of taxing all of them
    t_leg = leg("settling", t_leg)
    prof_stride = max(1, (n_regions * args.steps) // 10)
    eng.profile_enable(128, stride=prof_stride)  # at most ~6 launches per step and site: <= ~60 timed launches each
    region_s = []
    coll_before = n_coll[0]
    for reg in range(n_regions):
        first = reg * args.steps
        sync()
        t0 = time.perf_counter()
        for it in range(args.steps):
            step(ring[first + it])
        eng.flush_diagnostics()
        if world > 1:
            all_reduce(ring[first:first + args.steps])
        sync()
        region_s.append(time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor(region_s, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # per region: the slowest rank
        region_s = [float(v) for v in t.cpu()]
    elapsed = float(np.median(region_s))
    t_leg = leg("timed_regions", t_leg)
    coll_per_step = (n_coll[0] - coll_before - n_regions) / max(1, n_regions * args.steps)  # without the ring reductions

    # the replicated factor must be bit-identical on every rank (a divergent C would silently corrupt the fit)
    c_identical = None
    if world > 1:
        bits = eng.C.view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * torch.arange(1, bits.numel() + 1, device=device).view_as(bits)).sum()])
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        c_identical = bool(torch.equal(lo_, hi_))

    # live roofline (HIP events inside the library, same stream as the kernels): EVERY hot launch site of the step is
    # timed; the block names the site that takes the largest share of the step (launches per step x average duration)
    from matcouply_amd import _engine as E

    N_loc, K, r = X.shape[0], cfg["K"], cfg["r"]
    N_tot = N_loc
    if world > 1:
        t = torch.tensor([N_loc], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        N_tot = int(t.item())
    S_X, S_B = 4.0 * N_loc * K, 4.0 * N_loc * r
    n_B, n_C = len(cfg["regs"][1]), len(cfg["regs"][2])
    NBk = (r + 15) // 16
    NBk = 4 if NBk == 3 else NBk
    xc_fused = "GRAM=" in eng.kernel_variant(E.PROF_XC) and "GRAM=0" not in eng.kernel_variant(E.PROF_XC)
    swept_now = eng.kernel_variant(E.PROF_SWEEP) != "" and eng.profile_launches(E.PROF_SWEEP) > 0
    # the partial images the sweep leaves for the C- and A-phase (one per bseg, or per group of bsegs of a slab)
    n_parts, part_floats = 0, 0
    if swept_now:
        pb = eng.internal(E.BUF_BSEG_PART).view(torch.int32).cpu().numpy().astype(np.int64)
        n_parts = int((pb & 0x0FFFFFFF).max()) + 1 if len(pb) else 0
        kc = 2 if (K <= 128 and NBk == 1) else 4 * ((K + 255) // 256)
        part_floats = kc * 64 * 16 * NBk
    W16 = 16 * NBk
    n_tiles_B = int(sum((int(row_ptr[i + 1] - row_ptr[i]) + 63) // 64 for i in range(len(row_ptr) - 1)))
    # ALGORITHMIC bytes per launch of every site (reads + writes the design needs):
    #   X passes: X once (+ B-sized operands); fused rows / chained passes: their B-sized streams; the sweep: X once, aux / dual
    #   in, B / aux / dual out (X C never reaches memory); the [G | R] reduction and the A-phase finish: the partial images
    #   of the sweep (fp32 M, a-weighted Gram / fp64 B^T B); the C-phase finish: [G | R] in fp64, C and its ADMM variables;
    #   unimodal regressions: y in, fit out, once more for the split search = 16 B per element; PARAFAC2 algebra: the per-tile
    #   Gram statistics in, T_i out
    alg_bytes = {
        E.PROF_XC: S_X + (2 if xc_fused else 1) * S_B, E.PROF_XT: S_X + S_B, E.PROF_ROWS_FUSED: (2 + 4 * n_B) * S_B,
        E.PROF_SWEEP: S_X + (1 + 4 * n_B) * S_B,
        E.PROF_REDUCE: (4.0 * n_parts * (part_floats + W16 * W16 + W16) if swept_now else None),
        E.PROF_C_FINISH: 8.0 * (K * r + r * r) + (1 + 4 * n_C) * 4.0 * K * r + 4.0 * K * W16,
        E.PROF_A_FINISH: (4.0 * n_parts * part_floats + 8.0 * n_parts * r * r + 4.0 * K * W16) if swept_now else None,
        E.PROF_ROWS_CHAIN: (3 + 2 * n_B) * S_B, E.PROF_UNIMODAL: 4 * S_B,
        E.PROF_PF2: 8.0 * n_tiles_B * W16 * W16 + 12.0 * I_loc * r * r,
    }
    bound_of = {E.PROF_REDUCE: "latency (hbm figure for reference)", E.PROF_C_FINISH: "latency (hbm figure for reference)",
                E.PROF_A_FINISH: "latency (hbm figure for reference)", E.PROF_PF2: "fp64 MFMA issue (hbm figure for reference)",
                E.PROF_UNIMODAL: "latency of the pooling chain, then hbm"}
    overhead_us = eng.profile_overhead_us()
    n_steps_timed = n_regions * args.steps

    def pmc_table():
        """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if a profile of this configuration
        exists; counters are collected in separate runs, never inside this one."""
        if world != 1:
            return None, None
        for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
            path = os.path.join(REPO, "profiles", f"{rnd}_{args.config}_pmc_traffic.json")
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)["kernels"], os.path.relpath(path, REPO)
        return None, None

    pmc_kernels, pmc_source = pmc_table()

    def achievable_gbps():
        """the read bandwidth THIS box delivers to a pure streaming kernel (mcl_read_bandwidth), measured in this process after
        the timed regions: over up to 2 GiB of X when X exceeds the 256 MB last-level cache, else over a 1 GiB scratch buffer"""
        try:
            n_bytes = X.numel() * 4
            if n_bytes >= (600 << 20):
                buf = X.view(-1)[: min(X.numel(), (2 << 30) // 4)]
            else:
                buf = torch.empty((1 << 30) // 4, dtype=torch.float32, device=device).zero_()
            return E.read_bandwidth(buf, repeats=10)
        except Exception as exc:  # a measurement aid: its failure must not cost the line
            print(f"[bench] read-bandwidth probe failed: {exc}", file=sys.stderr)
            return None

    hbm_achievable = achievable_gbps() if rank == 0 else None

    def pmc_traffic(kernel_variant):
        if not pmc_kernels:
            return None
        base = kernel_variant.split("<")[0].split(" ")[0]
        # a site that launches several kernel forms (chained row passes): the launch-weighted mean of its kernels
        hits = [v for name, v in pmc_kernels.items() if name.split("<")[0] == base or
                (base == "k_rows_finish_solve_stats" and name.split("<")[0] in ("k_rows_solve_stats", "k_rows_finish_fused")) or
                (base == "k_rows_chain_first" and name.split("<")[0].startswith("k_rows_chain_"))]
        if not hits:
            return None
        n = sum(v["launches"] for v in hits)
        return int(sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in hits) / max(n, 1))

    per_kernel = []
    for slot in range(E.PROF_SLOTS):
        tot_ms, n = eng.profile_read(slot)
        launches = eng.profile_launches(slot)
        if not n or not launches:
            continue
        avg_us = 1e3 * tot_ms / n
        # the event pair itself costs `overhead_us` of marker handling (calibrated on an empty pair): the kernel's share
        net_us = max(avg_us - overhead_us, 0.0)
        ab = alg_bytes.get(slot)
        per_step = launches / max(1, n_steps_timed)
        e = dict(role=E.PROF_ROLE[slot], kernel=eng.kernel_variant(slot), launches_per_step=round(per_step, 3),
                 avg_us=round(net_us, 2), avg_us_with_event_pair=round(avg_us, 2), launches_timed=n,
                 us_per_step=round(per_step * net_us, 2),
                 algorithmic_bytes=int(ab) if ab else None,
                 achieved_gbps=round(ab / (net_us * 1e-6) / 1e9, 1) if ab and net_us > 0 else None,
                 frac=round(ab / (net_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if ab and net_us > 0 else None,
                 frac_achievable=(round(ab / (net_us * 1e-6) / 1e9 / hbm_achievable, 4) if ab and net_us > 0 and hbm_achievable else None),
                 traffic=pmc_traffic(eng.kernel_variant(slot)), bound=bound_of.get(slot, "hbm"), slot=slot)
        per_kernel.append(e)
    per_kernel.sort(key=lambda e: -e["us_per_step"])

    def mfma_block(slot, avg_ms):
        """The kernel's second limiter: fp32 MFMA flops it ISSUES per launch (padded rank / K as the tiles are) over the same
        HIP-event time, against the dense fp32 matrix peak.  At config 3 the analytic count (10.47 GFLOP) equals
        SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 of profiles/r2_c3_sq_counters.json; fp32 MFMA and VALU work do not overlap on a
        SIMD (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so the sweep sits at the SUM of this and its VALU / LDS / wait time."""
        RPk = 16 * ((r + 15) // 16)
        Kp = 128 if (slot == E.PROF_SWEEP and K <= 128 and RPk == 16) else 256 * ((K + 255) // 256) if slot == E.PROF_SWEEP else 64 * ((K + 63) // 64)
        n_inner = 5
        flops = {E.PROF_XC: 2.0 * N_loc * Kp * RPk + (2.0 * N_loc * RPk * RPk if xc_fused else 0.0),
                 E.PROF_XT: 2.0 * N_loc * Kp * RPk + 2.0 * N_loc * RPk * RPk,
                 E.PROF_ROWS_FUSED: 2.0 * N_loc * RPk * RPk * n_inner,
                 E.PROF_SWEEP: 4.0 * N_loc * Kp * RPk + 2.0 * N_loc * RPk * RPk * (n_inner + 2)}.get(slot)
        if flops is None or avg_ms <= 0:
            return None
        tf = flops / (avg_ms * 1e-3) / 1e12
        return dict(achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                    flops_per_launch=int(flops), counted="issued fp32 MFMA flops (analytic, padded tiles)")

    roofline = None
    timed_sites = [e for e in per_kernel if e["algorithmic_bytes"]]
    if timed_sites:
        dom = timed_sites[0]  # the largest share of the step among the sites with an algorithmic byte count
        slot = dom["slot"]
        roofline = dict(bound="hbm", achieved=dom["achieved_gbps"], peak=HBM_PEAK_GBS, unit="GB/s", frac=dom["frac"],
                        peak_achievable=round(hbm_achievable, 1) if hbm_achievable else None,
                        frac_achievable=dom["frac_achievable"],
                        peak_achievable_is=("streaming-read GB/s measured in this process by mcl_read_bandwidth (the data-sheet "
                                            "8 TB/s is `peak`)"),
                        traffic=dom["traffic"],
                        traffic_source=("static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command in separate "
                                        "passes, not measured in this run)" % pmc_source) if dom["traffic"] is not None else None,
                        kernel=dom["kernel"], kernel_role=dom["role"], launches_per_step=dom["launches_per_step"],
                        avg_us=dom["avg_us"], avg_us_with_event_pair=dom["avg_us_with_event_pair"],
                        event_pair_overhead_us=round(overhead_us, 2), launches_timed=dom["launches_timed"],
                        launches_in_timed_regions=int(round(dom["launches_per_step"] * n_steps_timed)), event_stride=prof_stride,
                        algorithmic_bytes_per_launch=dom["algorithmic_bytes"],
                        chosen_by="largest launches_per_step x avg_us of the step (per_kernel is sorted by it)",
                        note=dom["bound"] if dom["bound"] != "hbm" else None,
                        mfma=mfma_block(slot, dom["avg_us"] * 1e-3),
                        per_kernel=[{k: v for k, v in e.items() if k != "slot"} for e in per_kernel],
                        kernels_us_per_step=round(sum(e["us_per_step"] for e in per_kernel), 2))

    final = ring[n_regions * args.steps - 1].cpu().numpy() if args.steps else None
    if rank == 0:
        its = args.steps / elapsed
        S_X_tot, S_B_tot = 4.0 * N_tot * K, 4.0 * N_tot * r
        swept = swept_now
        bytes_iter = (S_X_tot + (1 + 4 * n_B) * S_B_tot) if swept else (2 * S_X_tot + (5 + 4 * n_B) * S_B_tot)
        out = {
            "metric": "AO-ADMM outer-iters/sec", "value": round(its, 2), "unit": "outer-iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 6),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["desc"], "I": cfg["I"], "J": cfg["J"], "sum_J": N_tot, "K": K, "rank": r,
                       "inner_n_iter_max": 5, "diagnostics_every_iteration": True,
                       "sharding": f"{world} x contiguous slab ranges" if world > 1 else "single device"},
            "algorithmic_bytes_per_iter": int(bytes_iter),
            "hbm_gbps_algorithmic": round(bytes_iter * its / 1e9, 1),
            "roofline": (dict(roofline, step_frac=round(bytes_iter / (1e-3 * 1e3 * elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                              step_frac_is="algorithmic_bytes_per_iter / ms_per_step / peak (whole step, all kernels and gaps)")
                         if roofline else None),
            "timed_regions": n_regions, "region_ms": [round(1e3 * v, 4) for v in region_s], "region_stat": "median",
            "region_ms_stats": {"min": round(1e3 * float(np.min(region_s)), 4), "median": round(1e3 * elapsed, 4),
                                "q25": round(1e3 * float(np.percentile(region_s, 25)), 4),
                                "q75": round(1e3 * float(np.percentile(region_s, 75)), 4),
                                "max": round(1e3 * float(np.max(region_s)), 4)},
            "settling": dict(settle, target_ms=args.settle_ms, rule="un-timed probe regions until >= target_ms and two "
                                                                    "consecutive probes within 2 %"),
        }
        if world > 1:
            out["collectives_per_step"] = round(coll_per_step, 2)
            out["collective_path"] = ("RCCL called directly on the engine's stream (matcouply_amd/_rccl.py)" if direct is not None
                                      else f"torch.distributed ({backend})")
            out["replicated_C_bit_identical"] = c_identical
        if final is not None:
            xsq, inner, model = final[5], final[3], final[4]
            out["final_rel_rec_error"] = round(float(np.sqrt(max(0.0, xsq - 2 * inner + model) / xsq)), 6)
        t_leg = leg("roofline+checks", t_leg)
        left = lambda: args.wall_budget - (time.perf_counter() - _T0)
        if world == 1 and not args.no_api:
            if left() > 0.5 * args.wall_budget:
                out["api"] = api_block(cfg, X, row_ptr)
            else:
                out["api"] = {"skipped": f"{time.perf_counter() - _T0:.0f} s of the --wall-budget of {args.wall_budget:.0f} s were "
                                         "used before this optional leg (wall_s names the slow leg)"}
            t_leg = leg("api_block", t_leg)
        if not args.no_cpu_baseline and world == 1:
            # the CPU leg is part of the contract line: never skipped, but its sample shrinks with the time that is left
            out["cpu_baseline"] = cpu_baseline(cfg, max(1.0, min(args.cpu_budget, 0.25 * left())))
            t_leg = leg("cpu_baseline", t_leg)
        elif world > 1:
            # the CPU leg is timed on rank 0 at N = 1 only (task contract); the N > 1 lines point at it
            out["cpu_baseline"] = {"value": None, "unit": "outer-iters/s", "cores": None, "kind": "port",
                                   "sample": "not timed at N > 1: see the cpu_baseline of the N = 1 line of the same workload"}
        WALL_S["total_until_print"] = round(time.perf_counter() - _T0, 3)
        out["wall_s"] = dict(WALL_S)
        print(json.dumps(out), flush=True)
    if world > 1:
        # the engine's own communicator (`direct`) is left to process exit: nothing more is sent on it, and tearing it down here
        # would add a collective step that can only delay or block the exit of a finished run
        dist.destroy_process_group()
    if rank == 0:
        print(f"[bench] done (t+{time.perf_counter() - _T0:.1f} s); interpreter teardown follows", file=sys.stderr, flush=True)


if __name__ == "__main__":
    sys.exit(main())
