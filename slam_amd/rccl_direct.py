"""RCCL called on the filter's own stream.

torch.distributed runs every collective on a stream of its own and ties it to the caller's stream with events; on this
hardware each such hand-over is about 10 us of idle GPU on either side of a collective that moves a few hundred bytes,
and a multi-rank filter step has five of them, each waiting for the one before.  This module calls the same library --
the librccl.so that torch ships -- through ctypes with the step's stream as the stream argument, so the collectives are
ordinary entries of that stream.  The communicator is a second one beside torch's: rank 0 creates the unique id, the
existing process group broadcasts it.

Opt-in (ShardedFilter(direct_rccl=True), or MCLHIP_RCCL_DIRECT=1 in the environment): it has run with a group of one
on the one-GPU test box, where it is bit-equal to the torch path; no run over several GPUs stands behind it yet.
"""
import ctypes as C
import os

_F64, _I64 = 8, 4          # ncclFloat64, ncclInt64
_MAX = 2                   # ncclMax


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


class RcclError(RuntimeError):
    pass


class DirectRccl:
    def __init__(self, torch, dist, rank, world):
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = lib = C.CDLL(path)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [i]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(vp), i, _UniqueId, i]
        lib.ncclCommDestroy.argtypes = [vp]
        lib.ncclAllReduce.argtypes = [vp, vp, sz, i, i, vp, vp]
        lib.ncclAllGather.argtypes = [vp, vp, sz, i, vp, vp]
        lib.ncclAllToAll.argtypes = [vp, vp, sz, i, vp, vp]
        for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclAllGather",
                     "ncclAllToAll"):
            getattr(lib, name).restype = i
        uid = _UniqueId()
        if rank == 0:
            self._ck(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            # (through the process group at hand; on the device where that group is RCCL itself)
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor(list(bytes(uid.internal)), dtype=torch.uint8, device=dev)
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().tolist())
            C.memmove(C.byref(uid), raw, 128)
        self.comm = vp()
        self._ck(lib.ncclCommInitRank(C.byref(self.comm), int(world), uid, int(rank)), "ncclCommInitRank")
        self.world = int(world)

    def _ck(self, rc, what):
        if rc != 0:
            raise RcclError("%s: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce_max_f64(self, t, stream):
        self._ck(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _F64, _MAX, self.comm, stream), "ncclAllReduce")

    def all_gather(self, out, inp, stream):
        """out [world * n] <- every rank's inp [n]; float64 or int64 (eight-byte elements either way)."""
        kind = _F64 if inp.dtype.is_floating_point else _I64
        if out.numel() != self.world * inp.numel() or out.element_size() != 8 or inp.element_size() != 8:
            raise RcclError("all_gather: shapes or element sizes do not match")
        self._ck(self.lib.ncclAllGather(inp.data_ptr(), out.data_ptr(), inp.numel(), kind, self.comm, stream), "ncclAllGather")

    def all_to_all_equal(self, recv, send, stream):
        """Block d of send goes to rank d, block s of recv comes from rank s; equal blocks of float64."""
        if recv.numel() != send.numel() or send.numel() % self.world or send.element_size() != 8:
            raise RcclError("all_to_all: blocks are not equal")
        self._ck(self.lib.ncclAllToAll(send.data_ptr(), recv.data_ptr(), send.numel() // self.world, _F64, self.comm, stream),
                 "ncclAllToAll")

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
