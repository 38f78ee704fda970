"""Host side of the multi-GPU communicator (``gsr_comm_*`` of include/gsr_hip.h, csrc/comm.hip).

``Comm.from_torch_group()`` builds the communicator a run uses: with the ``nccl`` backend (= RCCL over xGMI on ROCm) rank 0
asks the library for a unique id, ``torch.distributed`` broadcasts its 128 bytes ONCE, and every rank hands them to
``ncclCommInitRank`` inside the library -- from then on every collective of the data path is enqueued by the library on its own
stream; nothing crosses into Python per ICP iteration or per HEM level.  With ``gloo`` (CPU test boxes where several ranks
share the one GPU, which RCCL refuses) the same object is built over callbacks that bounce device buffers through host
memory -- the library calls them synchronously.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _lib

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

__all__ = ["Comm"]

_DT = {0: "float64", 1: "float32", 2: "int32", 3: "uint32", 4: "uint64"}


class _Ptr:
    """A raw device buffer as a torch tensor without a copy (``__cuda_array_interface__``)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 3}


def _view(ptr, nbytes, device):
    return torch.as_tensor(_Ptr(int(ptr), int(nbytes), "|u1"), device=torch.device("cuda", device))


class Comm:
    def __init__(self, handle, rank, world, device, keep=None, transport="rccl"):
        self._L = _lib.load(require_device=True)
        self._h = handle
        self.rank, self.world, self.device, self.transport = rank, world, device, transport
        self._keep = keep

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- constructors ------------------------------------------------------------------------------------------------
    @classmethod
    def rccl(cls, id_bytes: bytes, rank: int, world: int, device: int):
        """``ncclCommInitRank`` inside the library (collective: every rank must call it with the same id)."""
        L = _lib.load(require_device=True)
        assert len(id_bytes) == _lib.GSR_COMM_ID_BYTES
        buf = (C.c_char * _lib.GSR_COMM_ID_BYTES).from_buffer_copy(id_bytes)
        h = C.c_void_p()
        _lib.check(L.gsr_comm_create(C.byref(h), buf, int(rank), int(world), int(device)), "gsr_comm_create")
        return cls(h, rank, world, device, transport="rccl")

    @staticmethod
    def unique_id() -> bytes:
        L = _lib.load(require_device=True)
        buf = (C.c_char * _lib.GSR_COMM_ID_BYTES)()
        _lib.check(L.gsr_comm_get_unique_id(buf), "gsr_comm_get_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_group(cls, device: int, group=None, force_callbacks: bool = False):
        """The communicator of the default (or given) ``torch.distributed`` group; ``None`` for a single process."""
        if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # RCCL transport: the nccl backend, or -- ranks that share ONE GPU under gloo (test boxes) -- an explicitly named library
        # (GSR_RCCL_LIB: tests/mock_rccl, the test double of librccl for processes on one device) with GSR_COMM_TRANSPORT=rccl
        explicit = os.environ.get("GSR_COMM_TRANSPORT") == "rccl" and os.environ.get("GSR_RCCL_LIB")
        if (dist.get_backend(group) == "nccl" or explicit) and not force_callbacks:
            box = [cls.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            return cls.rccl(box[0], rank, world, device)
        return cls.callbacks_over_torch(rank, world, device, group)

    @classmethod
    def callbacks_over_torch(cls, rank, world, device, group=None):
        """The callback transport over ``torch.distributed`` host collectives (gloo): device buffers bounce through host memory."""
        L = _lib.load(require_device=True)

        def _allreduce(ptr, count, dtype, op, _user):
            try:
                name = _DT[int(dtype)]
                size = 8 if name in ("float64", "uint64") else 4
                raw = _view(ptr, int(count) * size, device)
                # gloo has no unsigned types: sums are taken on the signed view of the same bits (they wrap identically); an unsigned
                # 32-bit MAXIMUM is taken on zero-extended 64-bit values (the order-preserving codes of the bounding box have
                # their top bit set); an unsigned 64-bit maximum is not needed by the library
                tdt = {"float64": torch.float64, "float32": torch.float32, "int32": torch.int32, "uint32": torch.int32, "uint64": torch.int64}[name]
                host = raw.view(tdt).cpu()
                if int(op) == 0:
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                elif name == "uint32":
                    wide = host.to(torch.int64) & 0xFFFFFFFF
                    dist.all_reduce(wide, op=dist.ReduceOp.MAX, group=group)
                    host = wide.to(torch.int32)                                   # the low 32 bits back (the cast wraps)
                elif name == "uint64":
                    raise RuntimeError("unsigned 64-bit maximum is not supported by the callback transport")
                else:
                    dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
                raw.view(tdt).copy_(host)
                torch.cuda.synchronize(device)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def _allgather(send, recv, nbytes, _user):
            try:
                s = _view(send, nbytes, device).cpu()
                parts = [torch.empty(int(nbytes), dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, s, group=group)
                _view(recv, int(nbytes) * world, device).copy_(torch.cat(parts))
                torch.cuda.synchronize(device)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def _exchange(send, soff, sbytes, recv, roff, rbytes, _user):
            try:
                reqs, inbox = [], {}
                for r in range(world):
                    if r == rank:
                        continue
                    if sbytes[r] > 0:
                        reqs.append(dist.isend(_view(int(send) + soff[r], sbytes[r], device).cpu(), dst=r, group=group))
                    if rbytes[r] > 0:
                        inbox[r] = torch.empty(int(rbytes[r]), dtype=torch.uint8)
                        reqs.append(dist.irecv(inbox[r], src=r, group=group))
                for q in reqs:
                    q.wait()
                for r, t in inbox.items():
                    _view(int(recv) + roff[r], rbytes[r], device).copy_(t)
                torch.cuda.synchronize(device)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        cbs = _lib.CommCallbacks(_lib.COMM_ALLREDUCE_FN(_allreduce), _lib.COMM_ALLGATHER_FN(_allgather), _lib.COMM_EXCHANGE_FN(_exchange), None)
        h = C.c_void_p()
        _lib.check(L.gsr_comm_create_callbacks(C.byref(h), int(rank), int(world), int(device), C.byref(cbs)), "gsr_comm_create_callbacks")
        return cls(h, rank, world, device, keep=cbs, transport="callbacks")

    # ---- the operations, for tests and host code that shares the communicator ---------------------------------------------
    def all_reduce(self, t, op="sum", stream=None):
        code = {torch.float64: 0, torch.float32: 1, torch.int32: 2}[t.dtype]
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gsr_comm_allreduce(self._h, t.data_ptr(), t.numel(), code, 0 if op == "sum" else 1, C.c_void_p(st)), "gsr_comm_allreduce")

    def all_gather_bytes(self, send, recv, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.gsr_comm_allgather(self._h, send.data_ptr(), recv.data_ptr(), send.numel() * send.element_size(), C.c_void_p(st)), "gsr_comm_allgather")
