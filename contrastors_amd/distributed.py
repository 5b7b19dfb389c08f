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

    SPINS_PER_SECOND = 500_000   # measured: one poll of an uncached flag + s_sleep(8) is ~2 us
    MAX_SPINS = 60_000_000       # ~2 min of polling before a wait gives up (set_exchange_timeout / train_args.exchange_timeout_s):
                                 # long enough for a peer that stalls in its data loader or writes a checkpoint, short of RCCL's watchdog
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
        # Set-up is three phases so that a LOCAL failure (allocation, IPC export / open) can never leave the other ranks
        # blocked in a collective: every rank reaches every all_gather_object, carrying either its data or its error, and
        # all of them raise together.
        nb = self.N_BUF
        own, handles, err = [], None, None
        try:
            with torch.cuda.device(self.device):
                # receive buffers AND flags are uncached (fine-grained) device memory, as RCCL's own receive buffers are:
                # peers store into them over xGMI while this GPU's kernels read them, and only memory that bypasses the
                # reader's L2 is guaranteed to show remote stores without a stale line in between (a buffer is re-used
                # every N_BUF collectives).  The consumers stream through them once (clone / fixed-order sum).
                own = [self._alloc(self.cap, 1) for _ in range(nb)] + [self._alloc(256, 1), self._alloc(256, 2)]
                self._data, self._flags, self._err = own[:nb], own[nb], own[nb + 1]
                torch.as_tensor(_RawDeviceF32(own[nb], 64), device=self.device).zero_()
                C.memset(self._err, 0, 256)   # mapped host memory: the wait kernel stores here, check() reads it with no sync
                torch.cuda.synchronize(self.device)
                handles = [self._export(p) for p in own[: nb + 1]]
        except Exception as e:  # noqa: BLE001
            err = f"rank {self.rank}: {e}"
        everyone: List[Optional[object]] = [None] * self.world
        dist.all_gather_object(everyone, handles if err is None else RuntimeError(err), group=group)
        bad = [h for h in everyone if isinstance(h, Exception)]
        if not bad:
            try:
                with torch.cuda.device(self.device):
                    peers = [[] for _ in range(nb + 1)]   # data buffers, flags: one pointer per rank
                    for r, hs in enumerate(everyone):
                        for k in range(nb + 1):
                            peers[k].append(own[k] if r == self.rank else self._open(hs[k]))
                    as_dev = lambda ptrs: torch.tensor(ptrs, dtype=torch.int64, device=self.device)  # noqa: E731
                    self._peer_data = [as_dev(peers[k]) for k in range(nb)]
                    self._peer_flags = as_dev(peers[nb])
                    self._views = [torch.as_tensor(_RawDeviceF32(p, self.cap // 4), device=self.device) for p in self._data]
            except Exception as e:  # noqa: BLE001
                err = f"rank {self.rank}: {e}"
            oks: List[Optional[object]] = [None] * self.world
            dist.all_gather_object(oks, err, group=group)
            bad = [RuntimeError(o) for o in oks if o is not None]
        if bad:
            self._teardown_local()
            raise RuntimeError(f"one-shot exchange set-up failed ({bad[0]})")
        dist.barrier(group=group)   # nobody stores into a buffer its owner has not finished setting up

    def _teardown_local(self):
        for p in self._opened:
            self.lib.cx_ipc_close(p)
        self._opened = []
        for p in self._own:
            self.lib.cx_ipc_free(p)
        self._own = []

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


# ---- which path carries gather_with_grad: decided ONCE per process, by measurement -----------------
# mode "auto" (TrainArgs.exchange default; CX_EXCHANGE overrides): at the first exchange of fp32 CUDA embeddings every
# rank (1) sets the one-shot exchange up, (2) checks its all-gather and reduce-scatter bit-exact against the process
# group's collectives (RCCL on a real node) on integer-valued data, (3) times both at the call's own payload, and (4) takes
# the faster one -- the verdict is computed from MAX-over-ranks timings that are identical everywhere, and every step that
# can fail locally is followed by an agreement collective, so all ranks always choose the same path.  "rccl" skips all of
# it; "oneshot" keeps steps 1-2 and skips the race.  What happened is kept in exchange_report().
_ONESHOT: Optional[OneShotExchange] = None
_EXCHANGE_MODE = "auto"
_EXCHANGE_CHOICE: Optional[str] = None      # None = not decided yet; "oneshot" | "pg"
_EXCHANGE_REPORT: dict = {}


def set_exchange_mode(mode: str):
    """Config entry point (train_args.exchange): auto | rccl | oneshot.  Changing it re-opens the decision."""
    global _EXCHANGE_MODE, _EXCHANGE_CHOICE
    if mode not in ("auto", "rccl", "oneshot"):
        raise ValueError(f"exchange mode must be auto, rccl or oneshot, got {mode!r}")
    if mode != _EXCHANGE_MODE:
        _EXCHANGE_MODE, _EXCHANGE_CHOICE = mode, None


def set_exchange_timeout(seconds: float):
    """How long a rank's GPU polls for a peer's signal before the collective gives up (train_args.exchange_timeout_s)."""
    if not seconds > 0:
        raise ValueError(f"exchange timeout must be positive, got {seconds!r}")
    OneShotExchange.MAX_SPINS = max(1000, int(seconds * OneShotExchange.SPINS_PER_SECOND))


def exchange_mode() -> str:
    env = os.environ.get("CX_EXCHANGE")
    if env:
        if env not in ("auto", "rccl", "oneshot"):
            raise ValueError(f"CX_EXCHANGE must be auto, rccl or oneshot, got {env!r}")
        return env
    return _EXCHANGE_MODE


def exchange_report() -> dict:
    """What the selection measured and decided (bench.py prints it; the trainers log it once)."""
    return dict(_EXCHANGE_REPORT)


def check_exchange(sync: bool = False):
    """Surface a failed one-shot collective NOW rather than at the next one: the bounded wait of csrc/xgmi.hip gives up
    after its timeout (~2 min by default) when a peer never signals and reports through mapped host memory, which is only read at the start
    of the following collective.  The trainers call this once per optimizer step with sync=True -- one stream
    synchronisation per step (hundreds of milliseconds of work) so that a dead peer stops THIS step."""
    if _ONESHOT is None or _EXCHANGE_CHOICE != "oneshot":
        return
    if sync:
        torch.cuda.current_stream(_ONESHOT.device).synchronize()
    _ONESHOT.check()


def reset_exchange():
    """Forget the decision and release the one-shot buffers (tests; a re-initialised process group)."""
    global _ONESHOT, _EXCHANGE_CHOICE
    if _ONESHOT is not None:
        try:
            _ONESHOT.close()
        except Exception:  # noqa: BLE001
            pass
    _ONESHOT, _EXCHANGE_CHOICE = None, None
    _EXCHANGE_REPORT.clear()


def _agree(ok: bool, device) -> bool:
    """MIN over ranks of a flag: every rank takes the same branch."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _pg_all_gather(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out


def _pg_reduce_scatter(g: torch.Tensor, n: int) -> torch.Tensor:
    if dist.get_backend() == "nccl":
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM)
        return out
    # gloo (CPU tests, two processes sharing one GPU) has no reduce_scatter: all-reduce and keep our slice (same result)
    g = g.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    r = dist.get_rank()
    return g[r * n: (r + 1) * n].contiguous()


def _time_us(fn, iters: int = 5) -> float:
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def _select_exchange(shard: torch.Tensor) -> Optional[OneShotExchange]:
    """Run the selection described above for shards shaped like `shard` ((n, d) fp32 on this rank's GPU)."""
    global _ONESHOT, _EXCHANGE_CHOICE
    mode = exchange_mode()
    W, rank, dev = dist.get_world_size(), dist.get_rank(), shard.device
    rep = {"mode": mode, "backend": dist.get_backend(), "world": W, "shard_bytes": shard.numel() * 4}
    _EXCHANGE_REPORT.clear()
    _EXCHANGE_REPORT.update(rep)
    if mode == "rccl":
        _EXCHANGE_CHOICE = "pg"
        _EXCHANGE_REPORT.update(choice="pg", reason="configured")
        return None
    # (1) set-up: collective by construction (see OneShotExchange.__init__)
    try:
        if _ONESHOT is None or _ONESHOT.cap < W * shard.numel() * 4:
            if _ONESHOT is not None:
                _ONESHOT.close()
                _ONESHOT = None
            _ONESHOT = OneShotExchange(max(W * shard.numel() * 4, 64 << 20), device=dev)
        err = None
    except Exception as e:  # noqa: BLE001  (raised on every rank together)
        err = str(e)
    if err is not None:
        warnings.warn(f"one-shot xGMI exchange unavailable ({err}); using the process group's collectives")
        _ONESHOT, _EXCHANGE_CHOICE = None, "pg"
        _EXCHANGE_REPORT.update(choice="pg", reason=f"set-up failed: {err}")
        return None
    ex = _ONESHOT
    # (2) bit-exact check against the process group on integer-valued data (fp32 sums of small integers are exact in any
    # order, so RCCL's ring order and the one-shot path's rank order must agree to the last bit)
    n = shard.shape[0]
    rows = torch.arange(n * shard[0].numel(), device=dev, dtype=torch.float32).reshape(shard.shape) % 251
    probe = rows + 1000.0 * (rank + 1)
    gprobe = (torch.arange(W * n * shard[0].numel(), device=dev, dtype=torch.float32).reshape((W * n,) + tuple(shard.shape[1:])) % 127
              ) * (rank + 1)
    ok, why = True, ""
    try:
        a_ref, a_got = _pg_all_gather(probe), ex.all_gather(probe)
        r_ref, r_got = _pg_reduce_scatter(gprobe, n), ex.reduce_scatter(gprobe)
        torch.cuda.synchronize(dev)
        ex.check()
        if not torch.equal(a_ref, a_got):
            ok, why = False, "all-gather differs from the process group's"
        elif not torch.equal(r_ref, r_got):
            ok, why = False, "reduce-scatter differs from the process group's"
    except Exception as e:  # noqa: BLE001
        ok, why = False, str(e)
    if not _agree(ok, dev):
        warnings.warn(f"one-shot xGMI exchange failed its check against the process group ({why or 'on another rank'}); "
                      "using the process group's collectives")
        _EXCHANGE_CHOICE = "pg"
        _EXCHANGE_REPORT.update(choice="pg", verified=False, reason=f"verification failed: {why or 'on another rank'}")
        try:   #  the IPC buffers of an exchange that will never be used are given back
            ex.close()
        except Exception:  # noqa: BLE001
            pass
        _ONESHOT = None
        return None
    _EXCHANGE_REPORT.update(verified=True)
    if mode == "oneshot":
        _EXCHANGE_CHOICE = "oneshot"
        _EXCHANGE_REPORT.update(choice="oneshot", reason="configured")
        return ex
    # (3) + (4) the race, at this call's payload: all-gather + reduce-scatter of the step, MAX over ranks
    g_full = torch.zeros((W * n,) + tuple(shard.shape[1:]), dtype=torch.float32, device=dev)
    t_one = _time_us(lambda: (ex.all_gather(shard), ex.reduce_scatter(g_full)))
    t_pg = _time_us(lambda: (_pg_all_gather(shard), _pg_reduce_scatter(g_full, n)))
    t = torch.tensor([t_one, t_pg], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t_one, t_pg = float(t[0]), float(t[1])
    _EXCHANGE_CHOICE = "oneshot" if t_one < t_pg else "pg"
    _EXCHANGE_REPORT.update(choice=_EXCHANGE_CHOICE, oneshot_us=round(t_one, 1), pg_us=round(t_pg, 1),
                            reason="faster of the two on this node (all-gather + reduce-scatter at the step's payload)")
    if rank == 0:
        import logging

        logging.getLogger("contrastors_amd").info(
            f"embedding exchange: one-shot xGMI {t_one:.0f} us vs {dist.get_backend()} {t_pg:.0f} us per all-gather + "
            f"reduce-scatter of {W} x {shard.numel() * 4} B (verified bit-exact) -> {_EXCHANGE_CHOICE}")
    return ex if _EXCHANGE_CHOICE == "oneshot" else None


def _oneshot_for(shard: torch.Tensor) -> Optional[OneShotExchange]:
    """The one-shot exchange when it carries this process's gather_with_grad, else None (the process group does)."""
    global _ONESHOT, _EXCHANGE_CHOICE
    if _EXCHANGE_CHOICE is None:
        return _select_exchange(shard)
    if _EXCHANGE_CHOICE != "oneshot" or _ONESHOT is None:
        return None
    need = dist.get_world_size() * shard.numel() * 4
    if _ONESHOT.cap < need:   # a larger batch than the one the buffers were sized for: same on every rank -> collective
        #  the regrow can fail like the first set-up (IPC handle exchange, allocation): agree on the outcome across
        # ranks and fall back to the process group for the rest of the run instead of raising out of an autograd forward
        err = None
        try:
            _ONESHOT.close()
            _ONESHOT = None
            _ONESHOT = OneShotExchange(need, device=shard.device)
        except Exception as e:  # noqa: BLE001
            err = str(e)
        if not _agree(err is None, shard.device):
            warnings.warn(f"one-shot xGMI exchange could not be resized to {need} B ({err or 'failed on another rank'}); "
                          "using the process group's collectives from here on")
            try:
                if _ONESHOT is not None:
                    _ONESHOT.close()
            except Exception:  # noqa: BLE001
                pass
            _ONESHOT, _EXCHANGE_CHOICE = None, "pg"
            _EXCHANGE_REPORT.update(choice="pg", reason=f"resize failed: {err or 'on another rank'}")
            return None
    return _ONESHOT


class _AllGatherCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t: torch.Tensor) -> torch.Tensor:
        W = dist.get_world_size()
        t = t.contiguous()
        ctx.n = t.shape[0]
        ex = _oneshot_for(t) if (t.is_cuda and t.dtype == torch.float32 and (t.numel() * 4) % 16 == 0 and t.ndim >= 2) else None
        ctx.oneshot = ex is not None
        if ex is not None:
            return ex.all_gather(t)
        return _pg_all_gather(t)

    @staticmethod
    def backward(ctx, g: torch.Tensor) -> torch.Tensor:
        g = g.contiguous()
        if ctx.oneshot and _ONESHOT is not None and g.dtype == torch.float32:
            return _ONESHOT.reduce_scatter(g)
        return _pg_reduce_scatter(g, ctx.n)


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
