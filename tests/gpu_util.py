"""Thin raw-pointer wrappers over the C-ABI for the per-kernel GPU parity tests."""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np
import torch

from contrastors_amd import _C

OUT = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"


def report(name: str, **vals):
    """Append a diagnostics record (read back from gpurun_out/ after the GPU call)."""
    OUT.mkdir(exist_ok=True)
    rec = {"name": name}
    for k, v in vals.items():
        rec[k] = float(v) if isinstance(v, (int, float, np.floating)) or (torch.is_tensor(v) and v.ndim == 0) else v
    with open(OUT / "kernel_report.jsonl", "a") as f:
        f.write(json.dumps(rec) + "\n")


def S():
    return _C.cur_stream()


def L():
    return _C.lib()


def LD():
    """Development library: earlier GEMM generations, A/B attention kernels, probes, process-global switches."""
    return _C.dev_lib()


def bf(x):
    return x.to(torch.bfloat16).contiguous()


def gemm(x, w, bias=None, out_mode=0, split_k=1, out=None, alpha=1.0, lib=None):
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16 if out_mode == 0 else torch.float32, device=x.device)
        if out_mode == 2:
            out.zero_()
    _C.check((lib or L()).cx_gemm_bf16_nt(x.data_ptr(), w.data_ptr(), out.data_ptr(), _C.ptr(bias), M, N, K, x.stride(0),
                                 w.stride(0), out.stride(0), out_mode, split_k, alpha, S()), "gemm")
    return out


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())
