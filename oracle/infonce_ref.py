"""fp32 CPU restatement of the reference InfoNCE / GradCache math (TEST INFRASTRUCTURE -- see oracle/__init__.py).

  clip_loss        sc/loss.py:76-132   (labels :108-117, unidirectional x world_size :125, bidirectional :119-123)
  gather           sc/distributed.py:5-12 (rank-ordered concatenation; backward = sum over ranks of the slice grads)
  grad_cache_loss  sc/loss.py:187-213  (mathematically the full-batch loss; chunking only bounds memory)
  matryoshka step  sc/trainers/text_text.py:324-378 (direct step: un-normalised embeddings, one clip_loss per re-normalised
                   prefix, weighted sum; hard negatives are folded into the document side by the loader)
The multi-rank functions take the per-rank tensors of ALL ranks and emulate the collective in-process, so the
oracle needs no process group.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def labels_for(n_query: int, n_docs_all: int, rank: int, world: int) -> np.ndarray:
    """int64 label vector of sc/loss.py:108-117."""
    lab = np.arange(n_query, dtype=np.int64) + rank * n_query
    return lab * (n_docs_all // (n_query * world))


def clip_loss_ref(query: torch.Tensor, documents_all: torch.Tensor, scale: float, rank: int = 0, world: int = 1,
                  bidirectional: bool = False) -> torch.Tensor:
    """Loss of ONE rank given the already gathered documents."""
    labels = torch.from_numpy(labels_for(query.shape[0], documents_all.shape[0], rank, world)).to(query.device)
    sim = (query @ documents_all.T) * scale
    if bidirectional:
        sim_dq = (documents_all @ query.T) * scale
        return F.cross_entropy(sim, labels) + F.cross_entropy(sim_dq, labels)
    return F.cross_entropy(sim, labels) * world


def matryoshka_step_loss_ref(queries: torch.Tensor, documents_all: torch.Tensor, scale: float, dims=None, weights=None,
                             rank: int = 0, world: int = 1) -> torch.Tensor:
    """Loss of the direct (no GradCache) trainer step, sc/trainers/text_text.py:324-378, given what the model returned:
    with `dims` the embeddings arrive UN-normalised (normalize = matryoshka_dims is None, :325) and every prefix width is
    re-normalised before its own clip_loss (:352-369); without, the model normalised them and one clip_loss remains.
    Pinned to the reference's own function by tests/golden/matryoshka_step.npz."""
    if not dims:
        return clip_loss_ref(F.normalize(queries, dim=-1), F.normalize(documents_all, dim=-1), scale, rank, world)
    weights = list(weights) if weights is not None and len(weights) else [1.0] * len(dims)
    loss = 0.0
    for w, dim in zip(weights, dims):
        loss = loss + w * clip_loss_ref(F.normalize(queries[:, :dim], dim=-1), F.normalize(documents_all[:, :dim], dim=-1),
                                        scale, rank, world)
    return loss


def multi_rank_clip_loss_ref(queries: Sequence[torch.Tensor], documents: Sequence[torch.Tensor], scale: float):
    """Per-rank losses with gather_with_grad semantics; gradients flow to every rank's documents through the
    concatenation exactly as all_gather forward / reduce-scatter(SUM) backward does (one .backward() per rank's loss
    summed == each rank calling backward on its own loss)."""
    world = len(queries)
    docs_all = torch.cat(list(documents), dim=0)
    return [clip_loss_ref(q, docs_all, scale, r, world) for r, q in enumerate(queries)]


def infonce_rows_np(q: np.ndarray, d_all: np.ndarray, labels: np.ndarray, scale: float):
    """float64 numpy reference of the fused kernel's outputs: (lse_i, lse_i - logit_{i,label_i})."""
    s = (q.astype(np.float64) @ d_all.astype(np.float64).T) * scale
    m = s.max(axis=1, keepdims=True)
    lse = (m + np.log(np.exp(s - m).sum(axis=1, keepdims=True)))[:, 0]
    return lse, lse - s[np.arange(len(labels)), labels]
