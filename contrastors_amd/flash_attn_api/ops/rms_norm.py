"""flash_attn.ops.rms_norm (sc/layers/block.py:8, modeling_nomic_bert.py:18 -- `use_rms_norm`; False in all
five BASELINE configs): the LayerNorm kernels' RMS mode (cx_layernorm_fwd_mixed / _bwd_mixed, flag bit 4: no mean
subtraction, optional bias), same dtype handling and residual / prenorm / dropout conventions as dropout_add_layer_norm."""
import torch

from .layer_norm import _DropoutAddLN, dropout_add_layer_norm


def rms_norm(x, weight, epsilon):
    return _DropoutAddLN.apply(x, None, weight, None, epsilon, False, False, True)


def dropout_add_rms_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=None, layerscale=None, prenorm=False,
                         residual_in_fp32=False, return_dropout_mask=False):
    return dropout_add_layer_norm(x0, residual, weight, bias, dropout_p, epsilon, rowscale=rowscale, layerscale=layerscale,
                                  prenorm=prenorm, residual_in_fp32=residual_in_fp32,
                                  return_dropout_mask=return_dropout_mask, _rms=True)


def dropout_add_rms_norm_parallel_residual(*a, **k):
    raise NotImplementedError("ParallelBlock (GPT-J style) is decoder-only: out of scope (SURVEY.md §2b K7)")


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def forward(self, x):
        return rms_norm(x, self.weight, self.eps)
