"""A communicator of the engine's own: RCCL all-reduces enqueued DIRECTLY on the HIP stream the kernels run on.

`torch.distributed` runs every collective on a stream of its own and orders it against the caller's stream with two event
dependencies - about 8 us per collective on one MI355X (`tools/allreduce_overhead.py`), as much as the latency-bound
collectives of the sharded AO-ADMM loop themselves (35 KB of fp64 `[G | R]` per outer iteration, `r*r + 1` floats per
PARAFAC2 inner iteration: SURVEY.md 8e).  Here the same RCCL library torch has loaded is called through ctypes:
`ncclCommInitRank` with a unique id broadcast over the caller's process group, then `ncclAllReduce(..., stream)` with the
current HIP stream - the collective sits in the engine's stream between `k_reduce_frag` and the C-phase finish like any
kernel, no hand-over.  `DirectComm.try_create(group)` returns None (and the callers keep `torch.distributed`) when the
group is not an RCCL group, the library cannot be found, or the self-test against `torch.distributed` fails.
"""
import ctypes
import os


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]  # ncclUniqueId (nccl.h: NCCL_UNIQUE_ID_BYTES = 128)


_NCCL_SUM, _NCCL_MAX = 0, 2                  # ncclRedOp_t
_NCCL_FLOAT32, _NCCL_FLOAT64 = 7, 8          # ncclDataType_t
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    import torch

    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"]
    last = None
    for path in cands:  # torch's own copy first: the process then holds ONE RCCL
        try:
            lib = ctypes.CDLL(path)
            break
        except OSError as e:
            last = e
    else:
        raise OSError(f"librccl.so not found ({last})")
    P = ctypes.c_void_p
    lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = ctypes.c_int, [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = ctypes.c_int
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(P), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [P, P, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, P, P]
    lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = ctypes.c_int, [P]
    lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [ctypes.c_int]
    _lib = lib
    return lib


class DirectComm:
    """RCCL communicator over the ranks of a torch.distributed process group; collectives on torch's CURRENT stream."""

    # ncclCommInitRank is a collective bootstrap: bounded, so that a start-up problem costs time, not the run (on one node
    # it takes 1-3 s; try_create() makes every rank fall back to torch.distributed together when one of them times out)
    INIT_TIMEOUT_S = float(os.environ.get("MCL_RCCL_INIT_TIMEOUT_S", "30"))

    def __init__(self, group, device=None):
        import threading

        import torch
        import torch.distributed as dist

        self._torch, self.lib = torch, _load()
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.calls = 0
        self._comm = ctypes.c_void_p()
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # the communicator, its self-test and every later collective belong to ONE device: the engine's (the caller's data),
        # which need not be torch's current device when a host has not called set_device(local_rank)
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        dev = torch.device("cuda", dev_index)
        self.device = dev
        t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        # the bootstrap runs in a worker thread so that it can be abandoned: a rank that does not get its communicator in
        # time reports failure, try_create() then makes every rank fall back to torch.distributed together
        result = {}

        def init():
            try:
                torch.cuda.set_device(dev_index)
                comm = ctypes.c_void_p()
                result["rc"] = self.lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank)
                result["comm"] = comm
            except Exception as e:  # pragma: no cover
                result["error"] = e

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(self.INIT_TIMEOUT_S)
        if th.is_alive():
            raise TimeoutError(f"ncclCommInitRank did not return within {self.INIT_TIMEOUT_S:.0f} s")
        if "error" in result:
            raise result["error"]
        self._check(result["rc"], "ncclCommInitRank")
        self._comm = result["comm"]

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def all_reduce(self, t, op="sum"):
        torch = self._torch
        if not (t.is_cuda and t.is_contiguous()):
            raise ValueError("DirectComm.all_reduce needs a contiguous CUDA tensor")
        if t.device != self.device:
            raise ValueError(f"DirectComm of {self.device} asked to reduce a tensor on {t.device}")
        dt = {torch.float32: _NCCL_FLOAT32, torch.float64: _NCCL_FLOAT64}[t.dtype]
        stream = torch.cuda.current_stream(t.device).cuda_stream
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), dt, _NCCL_MAX if op == "max" else _NCCL_SUM,
                                           self._comm, ctypes.c_void_p(stream)), "ncclAllReduce")
        self.calls += 1

    def close(self):
        if self._comm:
            self.lib.ncclCommDestroy(self._comm)
            self._comm = ctypes.c_void_p()

    # no __del__: a communicator still alive at interpreter shutdown is left to process exit - ncclCommDestroy from a finaliser
    # (after torch.distributed or the HIP runtime have shut down, or while peers are already gone) can block or crash the exit
    # of an otherwise successful run; hosts that want it released call close() on every rank, after a barrier

    @classmethod
    def try_create(cls, group, device=None):
        """A working DirectComm for `group`, or None: only for RCCL ("nccl") groups, only if a self-test - an fp64 SUM and
        an fp32 MAX all-reduce compared with torch.distributed's - passes on EVERY rank; never raises."""
        import torch
        import torch.distributed as dist

        if os.environ.get("MCL_NO_DIRECT_RCCL") == "1":
            return None
        try:
            if dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
                return None
        except Exception:
            return None
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())

        def agreed(ok):
            """every rank takes the same decision: MIN over the ranks of the local verdict (a torch.distributed collective
            that EVERY rank reaches, whatever happened to it before)"""
            try:
                flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                return float(flag.item()) >= 1.0
            except Exception:
                return False

        # 1) the communicator (bounded bootstrap); no other collective until every rank has reported
        comm = None
        try:
            comm = cls(group, dev)
        except Exception:
            comm = None
        if not agreed(comm is not None):
            if comm is not None:
                comm._comm = ctypes.c_void_p()  # peers may never have joined: abandon it rather than destroy it collectively
            return None
        # 2) self-test against torch.distributed's result, again agreed by all
        ok = True
        try:
            g = torch.Generator(device=dev)
            g.manual_seed(1234 + comm.rank)
            a = torch.rand(4099, dtype=torch.float64, generator=g, device=dev)
            b = torch.rand(17, dtype=torch.float32, generator=g, device=dev)
            a_ref, b_ref = a.clone(), b.clone()
            dist.all_reduce(a_ref, group=group)
            dist.all_reduce(b_ref, op=dist.ReduceOp.MAX, group=group)
            comm.all_reduce(a)
            comm.all_reduce(b, "max")
            torch.cuda.synchronize(dev)
            # same ring / tree, same order of the sums inside RCCL: equal to rounding at worst, usually to the bit
            ok = bool(torch.allclose(a, a_ref, rtol=1e-13, atol=0.0) and torch.equal(b, b_ref))
        except Exception:
            ok = False
        if not agreed(ok):
            comm.close()
            return None
        comm.calls = 0
        return comm
