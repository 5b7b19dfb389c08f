"""The python symbol surface of `flash_attn` that contrastors imports, backed by
libcontrastors_hip.so.  `install()` registers this package under the name `flash_attn` so the reference's
`sc/layers/*.py` and `sc/models/encoder/modeling_nomic_bert.py` import it unchanged.

Supported on the device path: bf16 tensors, head_dim 64, non-causal; attention dropout (`dropout_p > 0`, the reference's
bert-base-uncased recipes) on the self-attention functions -- Philox masks keyed by the torch generator's (seed, offset), regenerated in
backward and under RandContext, never stored -- but not inside the kv-packed cross-attention.  That covers the five BASELINE configs;
anything else raises: there is no silent fallback to a generic implementation.
"""
from __future__ import annotations

import sys

from .flash_attn_interface import (flash_attn_kvpacked_func, flash_attn_qkvpacked_func,  # noqa: F401
                                   flash_attn_varlen_kvpacked_func, flash_attn_varlen_qkvpacked_func)

__version__ = "contrastors_amd-gfx950"


def install(name: str = "flash_attn") -> None:
    """Alias this package (and its sub-modules) as `flash_attn` in sys.modules."""
    import importlib

    me = sys.modules[__name__]
    sys.modules[name] = me
    for sub in ("flash_attn_interface", "bert_padding", "ops", "ops.layer_norm", "ops.rms_norm", "ops.fused_dense",
                "ops.activations", "layers", "layers.rotary", "losses", "losses.cross_entropy"):
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{sub}")
