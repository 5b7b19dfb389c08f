"""Load the REAL reference implementation by file path (build container only; /root/reference does not exist on the
GPU box).  Used solely by oracle/make_golden.py to produce tests/golden/*.npz.  TEST INFRASTRUCTURE.

`contrastors/__init__.py` pulls in flash_attn (absent), so a synthetic package object whose __path__ points at the
source tree is registered first; `wandb` (absent) is stubbed.  (SURVEY.md Appendix B recipe.)
"""
from __future__ import annotations

import importlib
import importlib.machinery as M
import importlib.util
import sys
import types
from pathlib import Path

REF_ROOT = Path("/root/reference/src/contrastors")


def available() -> bool:
    return (REF_ROOT / "loss.py").exists()


def load():
    """-> (ref_loss_module, ref_hf_config_module, ref_hf_model_module)"""
    if not available():
        raise RuntimeError("/root/reference is not present (only the build container has it)")
    if "contrastors" not in sys.modules:
        pkg = types.ModuleType("contrastors")
        pkg.__path__ = [str(REF_ROOT)]
        sys.modules["contrastors"] = pkg
    if "wandb" not in sys.modules:
        w = types.ModuleType("wandb")
        w.__spec__ = M.ModuleSpec("wandb", None)
        sys.modules["wandb"] = w
    ref_loss = importlib.import_module("contrastors.loss")
    base = REF_ROOT / "models" / "huggingface"
    if "refhf" not in sys.modules:
        p = types.ModuleType("refhf")
        p.__path__ = [str(base)]
        sys.modules["refhf"] = p

    def _load(n):
        full = f"refhf.{n}"
        if full in sys.modules:
            return sys.modules[full]
        spec = importlib.util.spec_from_file_location(full, str(base / f"{n}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        return m

    return ref_loss, _load("configuration_hf_nomic_bert"), _load("modeling_hf_nomic_bert")
