"""InfoNCE loss + GradCache on the fused HIP path (host-side mirror of sc/loss.py:76-213).

Public signatures are the reference's: `clip_loss(query, document, logit_scale, step, gather_enabled, tracker,
dataset, bidirectional)` and `grad_cache_loss(tower1, t1_inputs, tower2, t2_inputs, chunk_size, logit_scale,
bidirectional, router_aux_coeff)`.  What changes is underneath:
  * the (N x G) similarity + cross-entropy is one exact-fp32 MFMA kernel that never writes the logits
    (cx_infonce_fwd / cx_infonce_bwd), instead of matmul -> scale -> F.cross_entropy;
  * GradCache chunks run as single native calls over a pre-allocated arena; chunk boundaries / lengths are resolved
    on the host once per step (one sync per tower instead of one per model call);
  * the data-parallel gradient all-reduce is ONE flat collective per step issued here (the reference's DDP fires
    twice) -- towers expose `sync_gradients()`.
"""
from __future__ import annotations

import contextlib
import logging
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _C
from .distributed import gather_with_grad
from .policy import GradCachePolicy
from .rand_state import RandContext


def _scale_of(logit_scale) -> tuple:
    """(python float scale, parameter tensor or None).  LogitScale.forward(x) = x * exp(param)
    (sc/models/biencoder/modeling_biencoder.py:30-38)."""
    if isinstance(logit_scale, (int, float)):
        return float(logit_scale), None
    p = getattr(logit_scale, "logit_scale", None)
    if p is None:
        raise TypeError("logit_scale must be a LogitScale module or a number")
    const = getattr(logit_scale, "_const_scale", None)
    if const is not None and not p.requires_grad:
        return float(const), None
    return float(p.detach().exp().item()), (p if p.requires_grad else None)


class _FusedInfoNCEFp8(torch.autograd.Function):
    """Same loss with the similarity GEMM on the fp8 matrix cores (cx_infonce_fp8_fwd / _bwd; BASELINE configs[4],
    `use_fp8: true`).  Shapes the fp8 backward does not cover raise (no silent switch of numerics)."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, labels: torch.Tensor, scale: float, coef: float,
                scale_param: Optional[torch.Tensor]):
        if not q.is_cuda:
            raise RuntimeError("fused InfoNCE needs the HIP device path (no CPU fallback)")
        lib = _C.lib()
        q = q.float() if q.dtype != torch.float32 else q
        d = d.float() if d.dtype != torch.float32 else d
        q = q if q.stride(-1) == 1 else q.contiguous()
        d = d if d.stride(-1) == 1 else d.contiguous()
        N, dim = q.shape
        G = d.shape[0]
        if labels.shape[0] != N:
            raise ValueError(f"labels has {labels.shape[0]} entries for {N} query rows")
        if (N % 256) or (G % 64) or dim not in (256, 512, 768, 1024):
            raise ValueError(f"fp8 InfoNCE needs N % 256 == 0, G % 64 == 0, dim in (256, 512, 768, 1024); got {N}, {G}, {dim}")
        dev = q.device
        ws = torch.empty(lib.cx_infonce_fp8_ws_floats(N, G), dtype=torch.float32, device=dev)
        q8 = torch.empty(N, dim, dtype=torch.uint8, device=dev)
        d8 = torch.empty(G, dim, dtype=torch.uint8, device=dev)
        sq = torch.empty(N, dtype=torch.float32, device=dev)
        sd = torch.empty(G, dtype=torch.float32, device=dev)
        lse = torch.empty(N, dtype=torch.float32, device=dev)
        rows = torch.empty(N, dtype=torch.float32, device=dev)
        _C.check(lib.cx_infonce_fp8_fwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), scale, ws.data_ptr(), q8.data_ptr(),
                                        d8.data_ptr(), sq.data_ptr(), sd.data_ptr(), lse.data_ptr(), rows.data_ptr(), N, G,
                                        dim, q.stride(0), d.stride(0), _C.cur_stream()), "cx_infonce_fp8_fwd")
        ctx.save_for_backward(q, d, labels, lse, q8, d8, sq, sd)
        ctx.scale, ctx.coef, ctx.has_scale_param = scale, coef, scale_param is not None
        return rows.sum() * coef

    @staticmethod
    def backward(ctx, gout):
        q, d, labels, lse, q8, d8, sq, sd = ctx.saved_tensors
        lib = _C.lib()
        N, dim = q.shape
        G = d.shape[0]
        dev = q.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        gmt = torch.empty(G, N, **bf)
        qb, qbt, db = torch.empty(N, dim, **bf), torch.empty(dim, N, **bf), torch.empty(G, dim, **bf)
        gws = torch.empty(max(N * dim, 8 * N * dim), dtype=torch.float32, device=dev)
        dq = torch.empty(N, dim, dtype=torch.float32, device=dev)
        dd = torch.empty(G, dim, dtype=torch.float32, device=dev)
        dscale = torch.zeros(1, dtype=torch.float32, device=dev) if ctx.has_scale_param else None
        _C.check(lib.cx_infonce_fp8_bwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), lse.data_ptr(), ctx.scale, ctx.coef,
                                        q8.data_ptr(), d8.data_ptr(), sq.data_ptr(), sd.data_ptr(), gmt.data_ptr(),
                                        qb.data_ptr(), qbt.data_ptr(), db.data_ptr(), gws.data_ptr(), gws.numel(),
                                        dq.data_ptr(), dd.data_ptr(), _C.ptr(dscale), N, G, dim, q.stride(0), d.stride(0),
                                        _C.cur_stream()), "cx_infonce_fp8_bwd")
        gparam = (dscale[0] * ctx.scale * gout).reshape(()) if ctx.has_scale_param else None
        return dq * gout, dd * gout, None, None, None, gparam


def _infonce(q, d, labels, scale, coef, scale_param, use_fp8=False, argmax_out=None):
    """`use_fp8` (the recipe flag configs/train/contrastive_pretrain.yaml:24, TrainArgs.use_fp8) is an explicit argument
    all the way down: there is no process-wide switch.  argmax_out: see _FusedInfoNCE.forward (the fp8 kernel has no such
    output: its caller asks similarity_argmax)."""
    if use_fp8:
        return _FusedInfoNCEFp8.apply(q, d, labels, scale, coef, scale_param)
    return _FusedInfoNCE.apply(q, d, labels, scale, coef, scale_param, argmax_out)


def fp8_similarity_supported(n: int, g: int, dim: int) -> bool:
    """Shapes the fp8 similarity GEMM covers (what _FusedInfoNCEFp8 would otherwise refuse at the first step)."""
    return n % 256 == 0 and g % 64 == 0 and dim in (256, 512, 768, 1024)


class _FusedInfoNCE(torch.autograd.Function):
    """sum_i (lse_i - logit_{i,label_i}) * coef  for logits = scale * Q D^T, without materialising the logits."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, labels: torch.Tensor, scale: float, coef: float,
                scale_param: Optional[torch.Tensor], argmax_out: Optional[torch.Tensor] = None):
        """argmax_out ((N,) int32, optional): filled with every row's arg max over the G columns (cx_infonce_fwd_argmax) --
        the in-batch accuracy of sc/loss.py:127-130 without a second, materialised similarity matrix."""
        if not q.is_cuda:
            raise RuntimeError("fused InfoNCE needs the HIP device path (no CPU fallback)")
        lib = _C.lib()
        if q.dtype != torch.float32:
            q = q.float()
        if d.dtype != torch.float32:
            d = d.float()
        if q.stride(-1) != 1:
            q = q.contiguous()
        if d.stride(-1) != 1:
            d = d.contiguous()
        N, dim = q.shape
        G = d.shape[0]
        if labels.shape[0] != N:
            raise ValueError(f"labels has {labels.shape[0]} entries for {N} query rows")  # F.cross_entropy raises too
        if dim % 4:
            raise ValueError(f"fused InfoNCE needs dim % 4 == 0 (got {dim})")
        if ((N % 4) or (G % 4)) and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            # the backward's two output GEMMs contract over N and G: fail here, not in the middle of loss.backward()
            raise ValueError(f"fused InfoNCE backward needs N and G to be multiples of 4 (got {N}, {G})")
        lse = torch.empty(N, dtype=torch.float32, device=q.device)
        rows = torch.empty(N, dtype=torch.float32, device=q.device)
        if argmax_out is None:
            ws = torch.empty(lib.cx_infonce_ws_floats(N, G), dtype=torch.float32, device=q.device)
            _C.check(lib.cx_infonce_fwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), scale, ws.data_ptr(),
                                        lse.data_ptr(), rows.data_ptr(), N, G, dim, q.stride(0), d.stride(0),
                                        _C.cur_stream()), "cx_infonce_fwd")
        else:
            if argmax_out.dtype != torch.int32 or argmax_out.numel() != N or not argmax_out.is_contiguous():
                raise ValueError("argmax_out must be a contiguous (N,) int32 tensor")
            ws = torch.empty(lib.cx_infonce_argmax_ws_floats(N, G), dtype=torch.float32, device=q.device)
            _C.check(lib.cx_infonce_fwd_argmax(q.data_ptr(), d.data_ptr(), labels.data_ptr(), scale, ws.data_ptr(),
                                               lse.data_ptr(), rows.data_ptr(), argmax_out.data_ptr(), N, G, dim, q.stride(0),
                                               d.stride(0), _C.cur_stream()), "cx_infonce_fwd_argmax")
        ctx.save_for_backward(q, d, labels, lse)
        ctx.scale, ctx.coef, ctx.has_scale_param = scale, coef, scale_param is not None
        ctx.loss_rows = rows
        return rows.sum() * coef

    @staticmethod
    def backward(ctx, gout):
        q, d, labels, lse = ctx.saved_tensors
        lib = _C.lib()
        N, dim = q.shape
        G = d.shape[0]
        dev = q.device
        f32 = dict(dtype=torch.float32, device=dev)
        gm = torch.empty(N, G, **f32)
        gmt = torch.empty(G, N, **f32)
        qt = torch.empty(dim, N, **f32)
        dt = torch.empty(dim, G, **f32)
        dq = torch.empty(N, dim, **f32)
        dd = torch.empty(G, dim, **f32)
        dscale = torch.zeros(1, **f32) if ctx.has_scale_param else None
        _C.check(lib.cx_infonce_bwd(q.data_ptr(), d.data_ptr(), labels.data_ptr(), lse.data_ptr(), ctx.scale,
                                    ctx.coef, gm.data_ptr(), gmt.data_ptr(), qt.data_ptr(), dt.data_ptr(),
                                    dq.data_ptr(), dd.data_ptr(), _C.ptr(dscale), N, G, dim, q.stride(0),
                                    d.stride(0), _C.cur_stream()), "cx_infonce_bwd")
        dq = dq * gout
        dd = dd * gout
        # d loss / d log_scale = (d loss / d scale) * scale   (scale = exp(param))
        gparam = (dscale[0] * ctx.scale * gout).reshape(()) if ctx.has_scale_param else None
        return dq, dd, None, None, None, gparam, None


def similarity_argmax(query: torch.Tensor, document: torch.Tensor, labels: torch.Tensor, scale: float) -> torch.Tensor:
    """The arg max over dim 1 of scale * matmul(query, document.T), as (N,) int32, through the fused kernel (no (N, G) matrix, no vendor BLAS):
    for callers whose loss ran on another path (the fp8 similarity GEMM) and still want the reference's accuracy metric."""
    out = torch.empty(query.shape[0], dtype=torch.int32, device=query.device)
    with torch.no_grad():
        _FusedInfoNCE.apply(query.detach(), document.detach(), labels, float(scale), 1.0, None, out)
    return out


def make_labels(n_query: int, n_docs_all: int, rank: int, world: int, device) -> torch.Tensor:
    """int64 labels exactly as sc/loss.py:108-117: (arange(N) + rank*N) * (M_all // (N*world))."""
    labels = torch.arange(n_query, device=device)
    labels = labels + rank * n_query
    return labels * (n_docs_all // (n_query * world))


def clip_loss(query, document, logit_scale, step=None, gather_enabled=False, tracker=None, dataset="",
              bidirectional=False, *, use_fp8=False):
    """InfoNCE over (local queries) x (all gathered documents); see sc/loss.py:76-132 for the contract.

    One deliberate difference from the reference (INTEGRATION.md "quirks"): a label vector that would leave [0, G) -- e.g.
    `gather_enabled=False` in a multi-rank run, where sc/loss.py:108-117 silently produces labels pointing at other ranks'
    documents that are not there (integer-divide to 0) -- raises ValueError here instead of training on a wrong target."""
    if gather_enabled:
        document = gather_with_grad(document)
    if query.dtype != document.dtype:
        document = document.to(query.dtype)
    inited = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if inited else 0
    world = dist.get_world_size() if inited else 1  # the reference requires an initialised group (quirk 3)
    n = query.shape[0]
    G = document.shape[0]
    # the label vector is a host formula: check its range here (a label outside [0, G) would read a garbage logit;
    # F.cross_entropy in the reference raises) -- no device sync needed
    if n * world > G or (n - 1 + rank * n) * (G // (n * world)) >= G:
        raise ValueError(f"labels out of range: {n} queries on rank {rank}/{world} against {G} gathered documents")
    labels = make_labels(n, G, rank, world, query.device)
    scale, scale_param = _scale_of(logit_scale)
    if bidirectional and G != n:
        raise ValueError("bidirectional clip_loss needs as many documents as queries (sc/loss.py:119-123 only "
                         "type-checks for one process without negatives)")
    # sc/loss.py:127-130: with a tracker the in-batch accuracy (similarity.argmax(dim=1) == labels).mean() is logged.  The arg max
    # comes out of the loss kernel's own pass over the logit tiles (cx_infonce_fwd_argmax): no second similarity GEMM, no
    # materialised (n, G) matrix, no vendor BLAS (rounds 1-4 called torch.matmul here)
    argmax = torch.empty(n, dtype=torch.int32, device=query.device) if tracker is not None and not use_fp8 else None
    if bidirectional:
        # sc/loss.py:119-123: CE(q->d) + CE(d->q) with the same labels, no world-size factor
        l_qd = _infonce(query, document, labels, scale, 1.0 / n, scale_param, use_fp8, argmax)
        l_dq = _infonce(document, query, labels, scale, 1.0 / document.shape[0], scale_param, use_fp8)
        loss = l_qd + l_dq
    else:
        loss = _infonce(query, document, labels, scale, float(world) / n, scale_param, use_fp8, argmax)
    if tracker is not None:
        if argmax is None:   # the fp8 similarity kernel keeps no arg max: one more pass of the exact kernel, logits still unwritten
            argmax = similarity_argmax(query, document, labels, scale)
        acc = (argmax == labels).float().mean()
        tracker.log({f"accuracy/accuracy_{dataset}": acc.detach().cpu().item()}, step=step)
    return loss


# ----------------------------------------------------------------------------------------------------- GradCache
def _split_inputs(inputs: Dict[str, torch.Tensor], chunk_size: int, tail_seqs: int = 0) -> List[Dict]:
    """Chunk a tower's input dict along the batch axis and attach host-side sequence lengths (one sync per tower).
    tail_seqs (resident_tail_plan): the last tail_seqs sequences are chunked on their own, counted from the END, so that the
    kept region is whole chunks plus one shorter chunk at its front; the region before it is chunked as usual."""
    total = inputs["input_ids"].shape[0]
    lens = None
    mask = inputs.get("attention_mask")
    if "seqlens" in inputs:
        lens = inputs["seqlens"]
    elif mask is not None and mask.is_cuda:
        # right-padding check + lengths in one transfer
        lens_t = mask.sum(-1)
        right_padded = (mask[:, 1:] <= mask[:, :-1]).all()
        host = torch.stack([lens_t.max(), right_padded.to(lens_t.dtype)]).cpu()  # the one sync
        lens = lens_t.cpu().numpy() if bool(host[1]) else None
    tail_seqs = max(0, min(int(tail_seqs), total))
    head = total - tail_seqs
    bounds = [(a, min(a + chunk_size, head)) for a in range(0, head, chunk_size)]
    tail = []
    e = total
    while e > head:
        tail.append((max(head, e - chunk_size), e))
        e -= chunk_size
    bounds += tail[::-1]
    chunks = []
    for a, b in bounds:
        c = {k: v[a:b] for k, v in inputs.items() if torch.is_tensor(v) and v.shape[0] == total}
        if lens is not None:
            c["seqlens"] = lens[a:b]
        chunks.append(c)
    return chunks


def _uses_rng(model) -> bool:
    """Does a forward of this tower consume random numbers (dropout > 0 in training mode)?  Only then is the RNG snapshot
    of sc/loss.py:141-145 worth its cost; with p = 0 (every BASELINE config) nothing is drawn and nothing is saved."""
    trunk = getattr(model, "trunk", None)
    return bool(getattr(trunk, "uses_rng", False)) if trunk is not None else bool(getattr(model, "training", False))


def get_chunked_embeddings(model, chunks, rand_states=None, keep_tail: int = 0, kept: Optional[dict] = None):
    """Pass 1 (sc/loss.py:135-146): no-grad chunk forwards; returns (N,d) embeddings.  `rand_states` (a list) receives one
    RandContext snapshot per chunk, taken right before the chunk's forward (loss.py:141-143).
    keep_tail / kept (beyond the reference's signature; resident_tail_plan): the LAST keep_tail chunks run as saving forwards
    and their outputs (autograd holds the activation arenas) go to kept[chunk index] -- pass 2 back-propagates them without a
    re-forward.  A saving forward that runs out of memory turns itself and the rest of the tail into plain no-grad forwards."""
    embs = []
    needed = rand_states is not None and _uses_rng(model)
    n = len(chunks)
    trainable = kept is not None and keep_tail > 0 and model.training and not getattr(model, "frozen_trunk", False)
    first_kept = n - keep_tail if trainable else n
    trunk = getattr(model, "trunk", None)
    for i, c in enumerate(chunks):
        if rand_states is not None:
            rand_states.append(RandContext(c, needed=needed))
        if i >= first_kept:
            suspended = (trunk.selective_checkpointing_suspended() if hasattr(trunk, "selective_checkpointing_suspended")
                         else contextlib.nullcontext())
            outstanding = getattr(trunk, "_outstanding", None)
            try:
                with torch.enable_grad(), suspended:
                    out = model(**c)["embedding"]
                kept[i] = out
                embs.append(out.detach())
                continue
            except torch.OutOfMemoryError:
                if outstanding is not None:
                    # an allocation above the engine call (projection, hamming LayerNorm) failed after the saving forward was
                    # counted: its arena dies with the half-built graph, the count of saved forwards goes back too
                    trunk._outstanding = outstanding
                first_kept = n   # the estimate was too generous: what is kept so far stays, the rest is recomputed in pass 2
                #  the failed forward may have drawn its dropout offsets already: the retry below starts from the
                # snapshot pass 2 replays, not from wherever the failure left the generators
                if rand_states is not None:
                    rand_states[-1].restore()
                _log_once(("gradcache-tail-oom",), "GradCache: a kept chunk ran out of memory; the rest of the step re-forwards")
        with torch.no_grad():
            embs.append(model(**c)["embedding"])
    return torch.cat(embs, dim=0)


def accumulate_gradients(model, chunks, cache, rand_states=None, final: bool = False, skip=()):
    """Pass 2 (sc/loss.py:149-161): re-forward under the chunk's saved RNG state, back-propagate the cached embedding
    gradient through it.  `final`: this call's last chunk completes the tower's gradients for the step -- its backward
    starts the data-parallel reduction block by block (the reference's DDP does the same on the chunk it does not wrap in
    no_sync, sc/loss.py:151).  `skip`: chunk indices whose backward has already run (kept chunks, resident_tail_plan)."""
    todo = [i for i in range(len(chunks)) if i not in skip]
    for i in todo:
        c, g = chunks[i], cache[i]
        state = rand_states[i] if rand_states is not None else RandContext(c, needed=False)
        if final and i == todo[-1] and hasattr(model, "arm_overlapped_reduce"):
            model.arm_overlapped_reduce()
        with state:
            out = model(**c)["embedding"]
        # sc/loss.py:158-161 builds `surrogate = dot(reps.flatten(), gradient.flatten())` and back-propagates that scalar; its
        # gradient with respect to `reps` IS `gradient`, so the cached gradient is handed to autograd directly -- no vendor-BLAS
        # dot kernel (and no BLAS handle, whose workspace allocation was the first thing to fail under memory pressure)
        out.backward(g.to(out.dtype))


def cache_loss(query_embeddings, document_embeddings, logit_scale, bidirectional=False, *, use_fp8=False):
    """sc/loss.py:164-184: loss on detached embeddings, gradients w.r.t. the embeddings only."""
    q = query_embeddings.detach().requires_grad_()
    d = document_embeddings.detach().requires_grad_()
    loss = clip_loss(q, d, logit_scale, gather_enabled=True, bidirectional=bidirectional, use_fp8=use_fp8)
    loss.backward()
    return q.grad, d.grad, loss.detach()


_LOGGED: set = set()
LAST_SCHEDULE: dict = {}   # what the last grad_cache_loss call of this process ran (bench.py reports it per rank)


def _log_once(key, msg: str):
    if key in _LOGGED:
        return
    _LOGGED.add(key)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0:
        logging.getLogger("contrastors_amd").info(msg)


def effective_chunk(tower, inputs, chunk_size: int, policy: Optional[GradCachePolicy] = None) -> int:
    """The GradCache chunk is a pure memory knob: embeddings, loss and gradients do not depend on it (tested to fp32
    summation order).  The reference recipes say 64 because an 80 GB part cannot hold more activations; an MI355X has
    288 GB, and its GEMMs want >= 512 row panels per launch.  With policy.chunk == "auto" (the default) a recipe's
    chunk_size is therefore treated as a LOWER bound and raised -- in multiples of itself -- until a chunk carries ~262144
    tokens or its activation arena would take more than a third of the free HBM.  "exact" keeps the recipe's number; an
    integer forces it."""
    mode = (policy or GradCachePolicy()).with_env().chunk
    if mode == "exact" or chunk_size is None or chunk_size <= 0:
        return chunk_size
    if mode != "auto":
        return max(1, int(mode))
    cfg = getattr(getattr(tower, "trunk", None), "config", None)
    ids = inputs.get("input_ids") if isinstance(inputs, dict) else None
    if cfg is None or ids is None or ids.ndim != 2 or not ids.is_cuda:
        return chunk_size
    B, S = ids.shape
    bytes_per_token = _arena_bytes_per_token(tower)
    free, _ = torch.cuda.mem_get_info(ids.device)
    free += torch.cuda.memory_reserved(ids.device) - torch.cuda.memory_allocated(ids.device)  # the allocator's own cache
    free += sum(a.nbytes() for a in getattr(tower.trunk, "_arena_free", []))                  # arenas the engine re-uses
    tokens = int(min(262144, max(S, free / 3 / bytes_per_token)))
    want = max(chunk_size, tokens // max(S, 1))
    want = max(chunk_size, want // chunk_size * chunk_size)
    return int(min(want, max(B, chunk_size)))


def _arena_bytes_per_token(tower) -> float:
    """Saved activations + backward scratch of the native encoder, bytes per token (see nomic_bert._ChunkArena)."""
    cfg = tower.trunk.config
    d, I, L = cfg.n_embd, cfg.n_inner, cfg.n_layer
    wfc1 = 2 * I if getattr(cfg, "gated", False) else I
    per_layer = 2 * (3 * d + 5 * d + I + I) + 4 * (cfg.n_head + 4)   # (the gated MLP keeps the gate alone: (T, I), not (T, 2I))
    kept = 1 if getattr(tower.trunk, "gradient_checkpointing", False) else L
    needs_tr = any(f % 256 for f in (d, 3 * d, I, wfc1))   # (the transposed wgrad operands: nomic_bert._ChunkArena)
    scratch = 2 * (3 * d + (3 if needs_tr else 1) * max(3 * d, wfc1) + I)
    return kept * per_layer + (L - kept) * 2 * d + scratch


def resident_activations_fit(tower1, t1_inputs, tower2, t2_inputs, policy: Optional[GradCachePolicy] = None) -> bool:
    """GradCache exists because one pass cannot hold a big batch's activations (sc/loss.py:187-213 was written for 80 GB
    parts): pass 1 throws them away and pass 2 recomputes them, 4 forward-equivalents of FLOPs instead of 3.  When the
    WHOLE per-GPU batch of both towers fits in HBM (2048 pairs x 128 tokens of nomic-bert-2048: 193 GB of 288), pass 1
    can keep them and pass 2 has nothing to recompute: same embeddings, same loss, same gradients (the same kernels in the
    same order; identical up to the fp32-atomics noise two runs of the two-pass step show), a quarter of the work gone.
    policy.resident = "auto" | True | False; `auto` keeps the activations when they take <= 80 % of the free HBM (and
    grad_cache_loss falls back to the two-pass schedule if the estimate turns out wrong: torch.OutOfMemoryError there is
    caught before any gradient has been accumulated)."""
    mode = (policy or GradCachePolicy()).with_env().resident
    if mode is False:
        return False
    need = 0.0
    dev = None
    for tw, inp in ((tower1, t1_inputs), (tower2, t2_inputs)):
        ids = inp.get("input_ids") if isinstance(inp, dict) else None
        cfg = getattr(getattr(tw, "trunk", None), "config", None)
        if ids is None or cfg is None or not ids.is_cuda or ids.ndim != 2 or not hasattr(cfg, "n_inner"):
            return False
        if not tw.training or getattr(tw, "frozen_trunk", False):
            continue  # a frozen / eval tower saves nothing
        dev = ids.device
        mask = inp.get("attention_mask")
        tokens = ids.shape[0] * ids.shape[1] if mask is None else None
        if tokens is None:
            lens = inp.get("seqlens")
            tokens = int(sum(lens)) if lens is not None else ids.shape[0] * ids.shape[1]   # no sync: an upper bound is enough
        need += tokens * _arena_bytes_per_token(tw) * 1.03
    if dev is None:
        return False
    if mode is True:
        return True
    free = _agreed_free_bytes(dev, [tower1, tower2], agree=True)
    return need <= 0.8 * free


def _pool_free_bytes(dev, towers) -> float:
    free, _ = torch.cuda.mem_get_info(dev)
    free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)   # the allocator's own cache
    for tw in {id(t): t for t in towers}.values():
        free += sum(a.nbytes() for a in getattr(getattr(tw, "trunk", None), "_arena_free", []))   # arenas the engine re-uses
    return float(free)


# (device index, world size, id of the default process group) -> {"calls", "since", "free"}: the ranks' MIN of _pool_free_bytes.
# Keyed by the process group: a number agreed by one group must not survive into another trainer's group, a re-initialised
# group or a different world size.  reset_agreed_free() drops everything (trainer construction calls it).
_AGREED_FREE: Dict[tuple, dict] = {}
AGREED_REFRESH_EVERY = 64   # planner calls between two re-agreements (one scalar all-reduce each)


def _agreed_key(dev) -> tuple:
    grp = getattr(getattr(dist, "group", None), "WORLD", None)
    return (dev.index if dev.index is not None else -1, dist.get_world_size(), id(grp))


def reset_agreed_free() -> None:
    """Forget every agreed budget: the next planner call of each process group agrees again.  Must be called by ALL ranks at the same
    point of the program (trainer construction does); never from a rank-local event such as an out-of-memory fallback."""
    _AGREED_FREE.clear()


def note_local_memory_shortfall(dev, towers) -> None:
    """A planner-approved schedule ran out of memory on THIS rank (another tenant, fragmentation): until the group's next periodic
    re-agreement this rank plans with what it measures now if that is less -- no collective (the event is rank-local), and no repeat
    of the failed attempt on every following step (a stale-high agreed number made every step pay the attempt, the
    fallback's drop_idle_arenas + empty_cache and the two-pass step)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    st = _AGREED_FREE.get(_agreed_key(dev))
    if st is not None and st["free"] is not None:
        st["free"] = min(st["free"], _pool_free_bytes(dev, towers))


def _agreed_free_bytes(dev, towers, agree: bool = False) -> float:
    """What the memory planners below may spend.  One process: this device's free + pooled bytes, measured now.  Data parallel:
    the MINIMUM of that over the ranks, so that every rank plans the same schedule (all resident / the same kept tail / two
    passes) -- per-rank plans were correct (the collective counts match either way) but made the step time depend on the rank
    with the least free memory anyway, while the others re-forwarded less for nothing.  The agreement is ONE scalar all-reduce,
    issued only from call sites every rank reaches every step with rank-invariant conditions in front of it (`agree`:
    resident_activations_fit under policy "auto", resident_tail_plan when nothing has been agreed yet): in the first two such calls
    of a process group (the first sees the device before any arena exists) and then every AGREED_REFRESH_EVERY-th call -- the
    schedule is a deterministic function of the call count, so the ranks' collectives always pair up; memory that appears or
    disappears later (eval buffers, a second tenant, a different model) is seen within that many steps, not never.
    Between agreements the agreed number is reused: no per-step collective, no per-step host sync.  A rank whose memory shrinks
    in between is covered by the out-of-memory fallbacks, which also clamp ITS number (note_local_memory_shortfall)."""
    local = _pool_free_bytes(dev, towers)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return local
    st = _AGREED_FREE.setdefault(_agreed_key(dev), {"calls": 0, "since": 0, "free": None})
    if agree:
        due = st["calls"] < 2 or st["since"] >= AGREED_REFRESH_EVERY
        st["calls"] += 1
        st["since"] += 1
        if due:
            t = torch.tensor([local], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            st["free"] = float(t.item())
            st["since"] = 0
    return st["free"] if st["free"] is not None else local


def resident_tail_plan(tower1, t1_inputs, cq: int, tower2, t2_inputs, cd: int, policy: Optional[GradCachePolicy] = None):
    """(query sequences, document sequences) at the END of each side whose activations pass 1 keeps when the whole batch does
    not fit (resident_activations_fit said no): the two-pass schedule with as much of the re-forward removed as the HBM pays
    for.  The kept chunks are back-propagated FIRST in pass 2, so their arenas are back in the engine's pool when the
    re-forwards need a saving arena -- the budget is everything but the no-grad arena and the loss buffers: 85 % of (free +
    pooled) bytes.  Whole chunks from the end of the document side (then of the query side), plus a last shorter chunk in
    multiples of 64 sequences.  At the metric's single-GPU shape (2 x 16384 sequences of 128 tokens, 44 MB of saved
    activations each) that is ~5600 sequences: 17 % of pass 2's forward, ~4 % of the step; the same kernels on the same
    data, gradients accumulated in a different chunk order (fp32 summation order, as between any two chunk sizes).
    policy.resident: "auto" only (False: the reference's schedule literally; True asked for everything resident)."""
    mode = (policy or GradCachePolicy()).with_env().resident
    if mode != "auto":
        return 0, 0
    sides = []
    dev = None
    for tw, inp, chunk in ((tower1, t1_inputs, cq), (tower2, t2_inputs, cd)):
        cfg = getattr(getattr(tw, "trunk", None), "config", None)
        ids = inp.get("input_ids") if isinstance(inp, dict) else None
        ok = (cfg is not None and hasattr(cfg, "n_inner") and ids is not None and ids.is_cuda and ids.ndim == 2 and tw.training
              and not getattr(tw, "frozen_trunk", False) and hasattr(tw.trunk, "_arena_free") and chunk and chunk > 0)
        if not ok:
            sides.append(None)
            continue
        dev = ids.device
        lens = inp.get("seqlens")
        per_seq = (float(np.max(lens)) if lens is not None and len(lens) else float(ids.shape[1]))   # (an upper bound is enough)
        sides.append((ids.shape[0], int(chunk), per_seq * _arena_bytes_per_token(tw) * 1.03))
    if dev is None:
        return 0, 0
    towers = [t for t, sd_ in zip((tower1, tower2), sides) if sd_ is not None]
    # (data parallel: the ranks' minimum, normally agreed in resident_activations_fit one call earlier; a caller that comes here first
    # -- this function is reached under rank-invariant conditions too -- triggers the first agreement itself)
    nothing_agreed = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                      and _AGREED_FREE.get(_agreed_key(dev), {}).get("free") is None)
    budget = 0.85 * _agreed_free_bytes(dev, towers, agree=nothing_agreed)
    biggest = max(sd_[1] * sd_[2] for sd_ in sides if sd_ is not None)
    budget -= 0.09 * biggest + 3e9     # the no-grad arena (one slot instead of L: ~8 % of a saving arena) and the loss buffers
    keep = [0, 0]
    for side in (1, 0):                # document tail first, then the query tail
        if sides[side] is None:
            continue
        total, chunk, per_seq = sides[side]
        fit = int(max(0.0, budget) // per_seq)
        if fit >= total:
            keep[side] = total
            budget -= total * per_seq
            continue
        whole = fit // chunk * chunk
        part = (fit - whole) // 64 * 64      # one shorter chunk in front of the whole ones
        keep[side] = whole + (part if part >= 256 else 0)
        break
    return keep[0], keep[1]


def _resident_forward(model, chunks):
    """Forward of every chunk with its activations kept (autograd holds the arenas until the chunk's backward)."""
    if not model.training or getattr(model, "frozen_trunk", False):
        with torch.no_grad():
            return [model(**c)["embedding"] for c in chunks]
    outs = []
    trunk = getattr(model, "trunk", None)
    suspended = trunk.selective_checkpointing_suspended() if hasattr(trunk, "selective_checkpointing_suspended") else contextlib.nullcontext()
    try:
        with torch.enable_grad(), suspended:   # (one arena per chunk stays alive: see selective_checkpointing_suspended)
            for c in chunks:
                outs.append(model(**c)["embedding"])
    except torch.OutOfMemoryError:
        _release_resident(model, outs)   # the chunks that did fit: hand their arenas back before the caller falls back
        raise
    return outs


def _resident_backward(model, outs, cache, final: bool = False):
    live = [(o, g) for o, g in zip(outs, cache) if o.requires_grad]
    for i, (o, g) in enumerate(live):
        if final and i == len(live) - 1 and hasattr(model, "arm_overlapped_reduce"):
            model.arm_overlapped_reduce()
        o.backward(g.to(o.dtype))


def grad_cache_loss(tower1, t1_inputs, tower2, t2_inputs, chunk_size, logit_scale, bidirectional=False,
                    router_aux_coeff=False, *, policy: Optional[GradCachePolicy] = None):
    """GradCache step (sc/loss.py:187-213).  Leaves parameter gradients in the towers and returns the loss.  `policy`
    (keyword-only, beyond the reference's signature) carries the MI355X scheduling decisions; see GradCachePolicy."""
    pol = (policy or GradCachePolicy()).with_env()
    cq, cd = effective_chunk(tower1, t1_inputs, chunk_size, pol), effective_chunk(tower2, t2_inputs, chunk_size, pol)
    was_training1, was_training2 = tower1.training, tower2.training
    resident = resident_activations_fit(tower1, t1_inputs, tower2, t2_inputs, pol)
    # when the whole batch cannot stay resident, the tail of it that can is chunked on its own (resident_tail_plan)
    kq_seqs, kd_seqs = (0, 0) if resident else resident_tail_plan(tower1, t1_inputs, cq, tower2, t2_inputs, cd, pol)
    q_chunks = _split_inputs(t1_inputs, cq, kq_seqs)
    d_chunks = _split_inputs(t2_inputs, cd, kd_seqs)
    sizes_q = [c["input_ids"].shape[0] for c in q_chunks]
    sizes_d = [c["input_ids"].shape[0] for c in d_chunks]
    _log_once(("gradcache", chunk_size, cq, cd, resident, pol.use_fp8),
              f"GradCache schedule: recipe chunk_size {chunk_size} -> {cq} queries / {cd} documents per chunk "
              f"({len(q_chunks)} + {len(d_chunks)} chunks; policy.chunk = {pol.chunk!r}); "
              f"{'pass 1 keeps its activations, no re-forward' if resident else 'two passes (re-forward)'} "
              f"(policy.resident = {pol.resident!r}); similarity GEMM {'fp8' if pol.use_fp8 else 'fp32'}")
    done = False
    fell_back = False   # the all-resident attempt ran out of memory: this step takes the reference's schedule literally
    LAST_SCHEDULE.clear()
    LAST_SCHEDULE.update(schedule="resident" if resident else ("partial" if (kq_seqs or kd_seqs) else "two-pass"),
                         chunk_q=cq, chunk_d=cd, chunks_q=len(q_chunks), chunks_d=len(d_chunks), kept_q_seqs=kq_seqs,
                         kept_d_seqs=kd_seqs, fell_back=False)
    if resident:
        q_out = d_out = None
        try:
            # ONLY the forwards sit inside the try: they run before any collective, so a rank that falls back here issues
            # exactly the collectives its peers issue (one gather + one reduce-scatter inside cache_loss, one gradient
            # reduction).  An out-of-memory error inside cache_loss -- after gather_with_grad has run -- must propagate:
            # retrying it on this rank alone would issue one more all-gather than the peers and hang or desynchronise
            # the one-shot exchange's epochs.
            q_out = _resident_forward(tower1, q_chunks)
            d_out = _resident_forward(tower2, d_chunks)
        except torch.OutOfMemoryError:
            # the estimate of resident_activations_fit was wrong (fragmentation, another tenant of the allocator): no
            # parameter gradient has been touched yet (the encoder backward starts below), so drop what pass 1 kept,
            # hand the arenas back and take the two-pass schedule for this step.  policy.resident = True re-raises.
            _release_resident(tower1, q_out)
            _release_resident(tower2, d_out)
            q_out = d_out = None
            if pol.resident is True:
                raise
            fell_back = True
            LAST_SCHEDULE.update(schedule="two-pass", fell_back=True)
            # the arenas of the failed attempt went back to the engines' pools, not to the allocator: the device is still as
            # full as when the allocation failed, and the two-pass schedule needs a no-grad arena it does not have yet (round 5:
            # with a second tenant on the device the fallback itself ran out of memory).  Give everything idle back; the two
            # passes re-allocate the one saving arena and the one no-grad arena they need.
            for tw in {id(tower1): tower1, id(tower2): tower2}.values():
                drop = getattr(getattr(tw, "trunk", None), "drop_idle_arenas", None)
                if drop is not None:
                    drop()
            torch.cuda.empty_cache()
            ids_ = t1_inputs.get("input_ids") if isinstance(t1_inputs, dict) else None
            if ids_ is not None and ids_.is_cuda:
                note_local_memory_shortfall(ids_.device, [tower1, tower2])   # this rank stops re-trying until the next agreement
            _log_once(("gradcache-oom",), "GradCache: resident activations ran out of memory; falling back to the two-pass "
                                          "schedule (set train_args.gradcache_resident: false to skip the attempt)")
        else:
            try:
                q_cache, d_cache, loss = cache_loss(torch.cat([o.detach() for o in q_out]), torch.cat([o.detach() for o in d_out]),
                                                    logit_scale, bidirectional=bidirectional, use_fp8=pol.use_fp8)
            except torch.OutOfMemoryError:
                _release_resident(tower1, q_out)   # nothing leaks, and nothing is retried: the error is the caller's
                _release_resident(tower2, d_out)
                raise
            _resident_backward(tower1, q_out, q_cache.split(sizes_q), final=tower1 is not tower2 or not was_training2)
            _resident_backward(tower2, d_out, d_cache.split(sizes_d), final=True)
            del q_out, d_out
            done = True
    if not done:
        # two passes; under policy.resident = "auto" the tail of the batch that fits keeps its activations (resident_tail_plan)
        nq_keep = 0 if fell_back else (kq_seqs + cq - 1) // cq if kq_seqs else 0    # (chunks: the kept region was split on its own)
        nd_keep = 0 if fell_back else (kd_seqs + cd - 1) // cd if kd_seqs else 0
        if nq_keep or nd_keep:
            _log_once(("gradcache-tail", len(q_chunks), len(d_chunks), kq_seqs, kd_seqs),
                      f"GradCache schedule: the last {kq_seqs} query and {kd_seqs} document sequences ({nq_keep} + {nd_keep} chunks of "
                      f"{len(q_chunks)} + {len(d_chunks)}) keep their activations (no re-forward for them)")
        q_rnd, d_rnd = [], []
        q_kept, d_kept = {}, {}
        try:
            q_embs = get_chunked_embeddings(tower1, q_chunks, q_rnd, keep_tail=nq_keep, kept=q_kept)
            d_embs = get_chunked_embeddings(tower2, d_chunks, d_rnd, keep_tail=nd_keep, kept=d_kept)
            q_cache, d_cache, loss = cache_loss(q_embs, d_embs, logit_scale, bidirectional=bidirectional, use_fp8=pol.use_fp8)
        except BaseException:
            _release_resident(tower1, list(q_kept.values()))   # (nothing leaks when the loss, or a later forward, fails)
            _release_resident(tower2, list(d_kept.values()))
            raise
        q_grads, d_grads = q_cache.split(sizes_q), d_cache.split(sizes_d)
        done_q, done_d = set(q_kept), set(d_kept)       # (what pass 1 actually kept: an out-of-memory error may have cut the tail)
        re_q = len(done_q) < len(q_chunks)                                  # "the query side has chunks to re-forward"
        re_d = was_training2 and len(done_d) < len(d_chunks)
        # Kept chunks first: their arenas go back to the engine's pool and serve the re-forwards below.  The backward that
        # completes a tower's gradients starts their data-parallel reduction (arm_overlapped_reduce): the last re-forwarded
        # chunk where there is one (accumulate_gradients' `final`), else the last kept chunk of that tower.
        shared = tower1 is tower2
        for i in sorted(done_q):
            o = q_kept.pop(i)
            last_of_tower1 = not re_q and not q_kept and (not shared or (not done_d and not re_d))
            if last_of_tower1 and hasattr(tower1, "arm_overlapped_reduce"):
                tower1.arm_overlapped_reduce()
            o.backward(q_grads[i].to(o.dtype))
        for i in sorted(done_d):
            o = d_kept.pop(i)
            last_of_tower2 = not re_d and not d_kept and not (shared and re_q)
            if last_of_tower2 and hasattr(tower2, "arm_overlapped_reduce"):
                tower2.arm_overlapped_reduce()
            o.backward(d_grads[i].to(o.dtype))
        # (a tower shared by both sides has its gradients complete only after the document pass)
        if re_q:
            accumulate_gradients(tower1, q_chunks, q_grads, q_rnd, final=(not shared) or not re_d, skip=done_q)
        if re_d:
            accumulate_gradients(tower2, d_chunks, d_grads, d_rnd, final=True, skip=done_d)
    # data-parallel reduction of the accumulated gradients: once per step, one flat buffer per distinct tower
    seen = set()
    for tw, active in ((tower1, was_training1), (tower2, was_training2)):
        if active and id(tw) not in seen and hasattr(tw, "sync_gradients"):
            seen.add(id(tw))
            tw.sync_gradients()
    return loss


def _release_resident(tower, outs):
    """Give back the arenas the autograd graphs of a failed resident forward hold (see grad_cache_loss)."""
    if not outs:
        return
    for o in outs:
        fn = getattr(o, "grad_fn", None)   # the ctx of nomic_bert._EncodeFn / vit._VitEncodeFn when the tower is "plain"
        eng, arena = getattr(fn, "engine", None), getattr(fn, "arena", None)
        if eng is not None and arena is not None and hasattr(eng, "abandon_arena"):
            eng.abandon_arena(arena)
            fn.arena = None
