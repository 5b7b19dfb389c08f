"""Collectives of the InfoNCE path (host-side mirror of sc/distributed.py:5-29).

`gather_with_grad` is the one exchange step of the data-parallel loss: an all-gather of the
per-rank embeddings in RANK ORDER (labels depend on it, sc/loss.py:108-117) whose backward is a reduce-scatter(SUM)
of the gathered gradient.  On ROCm backend "nccl" is RCCL over xGMI; one fused buffer per call (not a list of W
tensors + torch.cat as the reference does) so each rank issues exactly one collective per direction.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import List, Optional

import torch
import torch.distributed as dist


def _world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class _RawDeviceF32:
    """torch view of device memory the library allocated (cx_ipc_alloc): torch.as_tensor reads __cuda_array_interface__."""

    def __init__(self, ptr: int, n_floats: int):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class OneShotExchange:
    """The loss path's exchange step in ONE hop over the fully connected xGMI fabric (csrc/xgmi.hip): each
    rank owns N_BUF receive buffers of `capacity_bytes` + a flag array, exported to the other per-GPU processes with HIP IPC;
    all_gather = every rank stores its shard into every peer's buffer + one flag exchange; reduce_scatter = the transpose
    + a local fixed-order sum.  fp32 only (the embeddings and their gradients).  Stream-ordered: no host synchronisation
    at all: the error flag of the bounded wait lives in mapped host memory and is polled at the start of every collective
    (a peer that never signals makes the wait give up; that surfaces as a RuntimeError at a later call instead of a hang)."""

    MAX_SPINS = 4_000_000   # ~8 s of polling before a wait gives up
    N_BUF = 4               # receive buffers, used round-robin: a result read in place stays valid for 3 more collectives

    def __init__(self, capacity_bytes: int, group=None, device: Optional[torch.device] = None):
        from . import _C

        self.lib = _C.lib()
        self._check = _C.check
        self._stream = _C.cur_stream
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device if device is not None else torch.cuda.current_device())
        self.cap = (int(capacity_bytes) + 255) // 256 * 256
        self.epoch = 0
        self._own: List[int] = []
        self._opened: List[int] = []
        with torch.cuda.device(self.device):
            nb = self.N_BUF
            own = [self._alloc(self.cap, 0) for _ in range(nb)] + [self._alloc(256, 1), self._alloc(256, 2)]
            self._data, self._flags, self._err = own[:nb], own[nb], own[nb + 1]
            torch.as_tensor(_RawDeviceF32(own[nb], 64), device=self.device).zero_()
            C.memset(self._err, 0, 256)   # mapped host memory: the wait kernel stores here, check() reads it with no sync
            torch.cuda.synchronize(self.device)
            handles = [self._export(p) for p in own[: nb + 1]]
            everyone: List[Optional[list]] = [None] * self.world
            dist.all_gather_object(everyone, handles, group=group)
            peers = [[] for _ in range(nb + 1)]   # data buffers, flags: one pointer per rank
            for r, hs in enumerate(everyone):
                for k in range(nb + 1):
                    peers[k].append(own[k] if r == self.rank else self._open(hs[k]))
            as_dev = lambda ptrs: torch.tensor(ptrs, dtype=torch.int64, device=self.device)  # noqa: E731
            self._peer_data = [as_dev(peers[k]) for k in range(nb)]
            self._peer_flags = as_dev(peers[nb])
            self._views = [torch.as_tensor(_RawDeviceF32(p, self.cap // 4), device=self.device) for p in self._data]
        dist.barrier(group=group)   # nobody stores into a buffer its owner has not finished setting up

    # ---- memory plumbing
    def _alloc(self, nbytes: int, uncached: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.cx_ipc_alloc(C.byref(p), nbytes, uncached), "cx_ipc_alloc")
        self._own.append(p.value)
        return p.value

    def _export(self, ptr: int) -> bytes:
        h = (C.c_ubyte * 64)()
        self._check(self.lib.cx_ipc_export(ptr, h), "cx_ipc_export")
        return bytes(h)

    def _open(self, handle: bytes) -> int:
        p = C.c_void_p()
        buf = (C.c_ubyte * 64).from_buffer_copy(handle)
        self._check(self.lib.cx_ipc_open(buf, C.byref(p)), "cx_ipc_open")
        self._opened.append(p.value)
        return p.value

    def close(self):
        torch.cuda.synchronize(self.device)
        for p in self._opened:
            self.lib.cx_ipc_close(p)
        self._opened = []
        if dist.is_initialized():
            dist.barrier(group=self.group)   # peers have unmapped before the owner frees
        for p in self._own:
            self.lib.cx_ipc_free(p)
        self._own = []

    def check(self):
        """Raise if a bounded wait of a collective that has already executed gave up (a plain read of mapped host memory:
        no synchronisation; a failure of work still queued is seen by a later call)."""
        e = C.c_uint32.from_address(self._err).value
        if e:
            raise RuntimeError(f"one-shot xGMI exchange: rank {self.rank} never received the signal of rank {e - 1}")

    def _begin(self, nbytes_total: int) -> int:
        if nbytes_total > self.cap:
            raise ValueError(f"exchange of {nbytes_total} bytes exceeds the {self.cap}-byte receive buffers")
        if self.epoch:
            self.check()
        self.epoch += 1
        return self.epoch % self.N_BUF

    def _signal_wait(self):
        self._check(self.lib.cx_xgmi_signal_wait(self._peer_flags.data_ptr(), self._flags, self.rank, self.world,
                                                 self.epoch & 0xFFFFFFFF, self.MAX_SPINS, self._err, self._stream()),
                    "cx_xgmi_signal_wait")

    # ---- collectives
    def all_gather(self, t: torch.Tensor, copy: bool = True) -> torch.Tensor:
        """(n, ...) fp32 -> (world * n, ...) in rank order.  copy=False returns a view of the receive buffer: valid until
        N_BUF - 1 further collectives have been issued on this exchange (peers write the buffer again after that)."""
        t = t.contiguous()
        assert t.dtype == torch.float32 and t.is_cuda
        nbytes = t.numel() * 4
        if nbytes % 16:
            raise ValueError("shard size must be a multiple of 16 bytes")
        b = self._begin(nbytes * self.world)
        self._check(self.lib.cx_xgmi_push(t.data_ptr(), self._peer_data[b].data_ptr(), self.rank * nbytes, nbytes, self.world,
                                          self._stream()), "cx_xgmi_push")
        self._signal_wait()
        out = self._views[b][: self.world * t.numel()]
        if copy:  # out of the receive buffer: autograd may hold the result longer than the buffer's validity window
            out = out.clone()
        return out.view((self.world * t.shape[0],) + tuple(t.shape[1:]))

    def reduce_scatter(self, g: torch.Tensor) -> torch.Tensor:
        """(world * n, ...) fp32 -> (n, ...): sum over ranks of their slice for this rank, in rank order (deterministic)."""
        g = g.contiguous()
        assert g.dtype == torch.float32 and g.is_cuda and g.shape[0] % self.world == 0
        n = g.shape[0] // self.world
        slice_elems = g.numel() // self.world
        if (slice_elems * 4) % 16:
            raise ValueError("slice size must be a multiple of 16 bytes")
        b = self._begin(slice_elems * 4 * self.world)
        self._check(self.lib.cx_xgmi_scatter(g.data_ptr(), self._peer_data[b].data_ptr(), self.rank, slice_elems * 4, self.world,
                                             self._stream()), "cx_xgmi_scatter")
        self._signal_wait()
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=torch.float32, device=g.device)
        self._check(self.lib.cx_sum_slots_f32(self._data[b], out.data_ptr(), slice_elems, self.world, self._stream()),
                    "cx_sum_slots_f32")
        return out


_ONESHOT: Optional[OneShotExchange] = None
_ONESHOT_BROKEN = False


def _oneshot_for(nbytes_total: int, device) -> Optional[OneShotExchange]:
    """CX_EXCHANGE=oneshot: the one-shot path for fp32 CUDA tensors (set up on first use; 64 MiB buffers cover 8 x 2048 x
    768 fp32 = 50 MB).  Any failure in the set-up disables it for the process and the RCCL collectives take over.  The
    default is RCCL: the one-shot path has run on 2 processes sharing one GPU (tests/test_distributed_gpu.py) but not yet
    on a multi-GPU node; `bench.py --gpus N` times both and says so in its xgmi_allgather record."""
    global _ONESHOT, _ONESHOT_BROKEN
    if os.environ.get("CX_EXCHANGE", "rccl") != "oneshot" or _ONESHOT_BROKEN:
        return None
    if _ONESHOT is None or _ONESHOT.cap < nbytes_total:
        try:
            if _ONESHOT is not None:
                _ONESHOT.close()
            _ONESHOT = OneShotExchange(max(nbytes_total, 64 << 20), device=device)
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"one-shot xGMI exchange unavailable ({e}); using the process group's collectives")
            _ONESHOT, _ONESHOT_BROKEN = None, True
            return None
    return _ONESHOT


class _AllGatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t: torch.Tensor) -> torch.Tensor:
        W = dist.get_world_size()
        t = t.contiguous()
        ctx.n = t.shape[0]
        ex = _oneshot_for(W * t.numel() * 4, t.device) if (t.is_cuda and t.dtype == torch.float32 and (t.numel() * 4) % 16 == 0) else None
        ctx.oneshot = ex is not None
        if ex is not None:
            return ex.all_gather(t)
        out = torch.empty((W * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor) -> torch.Tensor:
        g = g.contiguous()
        rank = dist.get_rank()
        if ctx.oneshot and _ONESHOT is not None and g.dtype == torch.float32:
            return _ONESHOT.reduce_scatter(g)
        if dist.get_backend() == "nccl":
            out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
            return out
        # gloo (CPU tests) has no reduce_scatter: all-reduce and keep our slice (same result)
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[rank * ctx.n: (rank + 1) * ctx.n].contiguous()


def gather_with_grad(t: torch.Tensor) -> torch.Tensor:
    """sc/distributed.py:5-12 -- identity for world size 1, else autograd-aware rank-ordered concatenation."""
    if _world() == 1:
        return t
    if t.ndim == 0:
        t = t.unsqueeze(0)
    return _AllGatherCat.apply(t)


def gather(t: torch.Tensor) -> torch.Tensor:
    """sc/distributed.py:15-29 -- no-grad all-gather; the local slice keeps its own tensor (and autograd history)."""
    if _world() == 1:
        return t
    if t.ndim == 0:
        t = t.unsqueeze(0)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    gathered[dist.get_rank()] = t
    return torch.cat(gathered, dim=0)


def print_rank_zero(*args, **kwargs):
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(*args, **kwargs)
