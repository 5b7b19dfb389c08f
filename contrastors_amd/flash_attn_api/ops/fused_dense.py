"""flash_attn.ops.fused_dense.FusedDense: K9 (sc/layers/attention.py:82-85, sc/layers/mlp.py:24-28,61-65).
nn.Linear-compatible parameters; forward/backward run the bf16 MFMA GEMM (fp32 accumulate, bf16 output; weight and
bias gradients in fp32).  The engine path (NomicBertEngine) keeps persistent bf16 / transposed shadows instead of the
per-call casts done here."""
from __future__ import annotations

import torch

from ... import _C


def _pad64(t: torch.Tensor) -> torch.Tensor:
    k = t.shape[-1]
    if k % 64 == 0:
        return t
    return torch.nn.functional.pad(t, (0, 64 - k % 64))


class _FusedDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _C.lib()
        shape = x.shape
        x2 = _pad64(x.reshape(-1, shape[-1]).to(torch.bfloat16)).contiguous()
        w16 = _pad64(weight.to(torch.bfloat16)).contiguous()
        M, K = x2.shape
        N = w16.shape[0]
        if N % 4:
            raise NotImplementedError("out_features must be a multiple of 4")
        out = torch.empty(M, N, dtype=torch.bfloat16, device=x.device)
        b32 = None if bias is None else bias.float().contiguous()
        _C.check(lib.cx_gemm_bf16_nt(x2.data_ptr(), w16.data_ptr(), out.data_ptr(), _C.ptr(b32), M, N, K, K, K, N, 0, 1,
                                     1.0, _C.cur_stream()), "gemm fwd")
        ctx.save_for_backward(x2, w16)
        ctx.meta = (shape, x.dtype, weight.dtype, bias is not None, weight.shape[1])
        return out.view(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dout):
        lib = _C.lib()
        x2, w16 = ctx.saved_tensors
        shape, xdtype, wdtype, has_bias, k_in = ctx.meta
        M, K = x2.shape
        N = w16.shape[0]
        s = _C.cur_stream()
        dy = dout.reshape(M, N).to(torch.bfloat16).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx = dy @ W : NT form needs W^T (K, N); N is the reduction -> pad to 64
            Np = (N + 63) // 64 * 64
            wt = torch.zeros(K, Np, dtype=torch.bfloat16, device=dy.device)
            wt[:, :N] = w16.t()
            dyp = dy if Np == N else torch.nn.functional.pad(dy, (0, Np - N))
            dxo = torch.empty(M, K, dtype=torch.bfloat16, device=dy.device)
            _C.check(lib.cx_gemm_bf16_nt(dyp.data_ptr(), wt.data_ptr(), dxo.data_ptr(), None, M, K, Np, Np, Np, K, 0, 1,
                                         1.0, s), "gemm dgrad")
            dx = dxo[:, :k_in].reshape(shape).to(xdtype)
        if ctx.needs_input_grad[1]:
            Mp = (M + 63) // 64 * 64
            Nn = (N + 7) // 8 * 8  # the transpose kernel moves 8 columns per lane
            dyn = dy if Nn == N else torch.nn.functional.pad(dy, (0, Nn - N)).contiguous()
            dyt = torch.empty(Nn, Mp, dtype=torch.bfloat16, device=dy.device)
            xt = torch.empty(K, Mp, dtype=torch.bfloat16, device=dy.device)
            _C.check(lib.cx_transpose_bf16(dyn.data_ptr(), dyt.data_ptr(), M, Nn, Nn, Mp, Mp, s), "transpose")
            _C.check(lib.cx_transpose_bf16(x2.data_ptr(), xt.data_ptr(), M, K, K, Mp, Mp, s), "transpose")
            g = torch.zeros(Nn, K, dtype=torch.float32, device=dy.device)
            ws = torch.empty(max(Nn * K, min(16 * Nn * K, 1 << 24)), dtype=torch.float32, device=dy.device)
            _C.check(lib.cx_gemm_bf16_nt_accum(dyt.data_ptr(), xt.data_ptr(), g.data_ptr(), ws.data_ptr(), ws.numel(), Nn,
                                               K, Mp, Mp, Mp, s), "gemm wgrad")
            dw = g[:N, :k_in].to(wdtype)
        if has_bias and ctx.needs_input_grad[2]:
            if N % 8 == 0:
                dbv = torch.zeros(N, dtype=torch.float32, device=dy.device)
                _C.check(lib.cx_bias_grad(dy.data_ptr(), dbv.data_ptr(), M, N, N, s), "bias grad")
            else:
                dbv = dy.float().sum(0)
            db = dbv
        return dx, dw, db


class FusedDense(torch.nn.Linear):
    def __init__(self, in_features, out_features, bias=True, return_residual=False, device=None, dtype=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        self.return_residual = return_residual

    def forward(self, x):
        out = _FusedDenseFn.apply(x, self.weight, self.bias)
        return out if not self.return_residual else (out, x)


def fused_dense_func(x, weight, bias=None, return_residual=False):
    out = _FusedDenseFn.apply(x, weight, bias)
    return out if not return_residual else (out, x)
