"""Scheduling decisions of the MI355X path as CONFIG: `TrainArgs.gradcache_chunk`,
`gradcache_resident`, `use_fp8`, `exchange` -> GradCachePolicy / distributed.set_exchange_mode.  The environment variables
of round 2 (CX_GRADCACHE_CHUNK, CX_GRADCACHE_RESIDENT, CX_EXCHANGE) remain as an operator OVERRIDE on top of the config.
No torch import here: config.py validates the fields when the YAML is read."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Union


@dataclass
class GradCachePolicy:
    """How grad_cache_loss schedules a step on a 288 GB part.  These are CONFIG fields (TrainArgs.gradcache_chunk /
    gradcache_resident / use_fp8, `GradCachePolicy.from_train_args`); the environment variables of round 2
    (CX_GRADCACHE_CHUNK, CX_GRADCACHE_RESIDENT) remain as an operator OVERRIDE on top of whatever the config says.
      chunk     "auto": the recipe's chunk_size is a lower bound, raised until a chunk carries ~262144 tokens or its arena
                would take a third of the free HBM | "exact": the recipe's number, literally | n: force n
      resident  "auto": keep pass 1's activations when they need <= 80 % of the free HBM (no re-forward) | True | False
      use_fp8   similarity GEMM of the loss on the fp8 matrix cores
    The schedule actually taken is logged once per distinct decision (logger "contrastors_amd")."""
    chunk: Union[str, int] = "auto"
    resident: Union[str, bool] = "auto"
    use_fp8: bool = False

    @classmethod
    def from_train_args(cls, ta) -> "GradCachePolicy":
        return cls(chunk=_parse_chunk(getattr(ta, "gradcache_chunk", "auto"), "train_args.gradcache_chunk"),
                   resident=_parse_resident(getattr(ta, "gradcache_resident", "auto"), "train_args.gradcache_resident"),
                   use_fp8=bool(getattr(ta, "use_fp8", False)))

    def with_env(self) -> "GradCachePolicy":
        c, r = os.environ.get("CX_GRADCACHE_CHUNK"), os.environ.get("CX_GRADCACHE_RESIDENT")
        return GradCachePolicy(chunk=self.chunk if c in (None, "") else _parse_chunk(c, "CX_GRADCACHE_CHUNK"),
                               resident=self.resident if r in (None, "") else _parse_resident(r, "CX_GRADCACHE_RESIDENT"),
                               use_fp8=self.use_fp8)


def _parse_chunk(v, where: str):
    if v is None or v == "":
        return "auto"
    if isinstance(v, str) and v.lower() in ("auto", "exact"):
        return v.lower()
    try:
        n = int(v)
    except (TypeError, ValueError):
        raise ValueError(f"{where} must be 'auto', 'exact' or a positive integer, got {v!r}") from None
    if n <= 0:
        raise ValueError(f"{where} must be 'auto', 'exact' or a positive integer, got {v!r}")
    return n


def _parse_resident(v, where: str):
    if v is None or v == "":
        return "auto"
    if isinstance(v, bool):
        return v
    t = str(v).lower()
    if t == "auto":
        return "auto"
    if t in ("1", "true", "yes", "on"):
        return True
    if t in ("0", "false", "no", "off"):
        return False
    raise ValueError(f"{where} must be 'auto', true / 1 or false / 0, got {v!r}")


def _parse_keep(v, where: str):
    """checkpoint_keep_layers (selective activation checkpointing): 'auto' or a non-negative integer; None -> 'auto'."""
    if v is None or v == "":
        return "auto"
    if isinstance(v, str) and v.strip().lower() == "auto":
        return "auto"
    if isinstance(v, bool):
        raise ValueError(f"{where} must be 'auto' or a non-negative integer, got {v!r}")
    try:
        n = int(v)
    except (TypeError, ValueError):
        raise ValueError(f"{where} must be 'auto' or a non-negative integer, got {v!r}") from None
    if n < 0:
        raise ValueError(f"{where} must be 'auto' or a non-negative integer, got {v!r}")
    return n
