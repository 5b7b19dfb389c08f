"""flash_attn.losses.cross_entropy.CrossEntropyLoss (K12; sc/models/encoder/modeling_nomic_bert.py:47,603-610 builds it
as `partial(CrossEntropyLoss, inplace_backward=True)` for the 30528-way MLM head) on the fused HIP kernel
cx_xent_fwd / cx_xent_bwd: one read of the logits forward, one read + one (optionally in-place) write backward.

Same constructor and call signature as the reference dependency.  Supported: ignore_index, reduction
{"mean","sum","none"}, logit_scale, inplace_backward; label_smoothing > 0, lse_square_scale > 0 (z-loss) and
tensor-parallel process groups raise.  No CPU path: non-GPU logits raise.
"""
import torch

from ... import _C


class _Xent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, logit_scale, ignore_index, inplace_backward):
        if not logits.is_cuda:
            raise RuntimeError("CrossEntropyLoss runs on the HIP kernel only (no CPU path)")
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise ValueError("logits must be (N, V) with unit stride along V")
        if logits.dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError(f"logits dtype {logits.dtype}")
        N, V = logits.shape
        labels = labels.to(torch.int64).contiguous()
        loss = torch.empty(N, dtype=torch.float32, device=logits.device)
        lse = torch.empty(N, dtype=torch.float32, device=logits.device)
        _C.check(_C.lib().cx_xent_fwd(logits.data_ptr(), int(logits.dtype == torch.bfloat16), labels.data_ptr(),
                                      loss.data_ptr(), lse.data_ptr(), N, V, logits.stride(0), float(logit_scale),
                                      int(ignore_index), _C.cur_stream()), "cx_xent_fwd")
        ctx.save_for_backward(logits, labels, lse)
        ctx.args = (float(logit_scale), int(ignore_index), bool(inplace_backward))
        ctx.mark_non_differentiable(lse)
        return loss, lse

    @staticmethod
    def backward(ctx, dloss, _dlse):
        logits, labels, lse = ctx.saved_tensors
        scale, ignore_index, inplace = ctx.args
        N, V = logits.shape
        dlogits = logits if inplace else torch.empty_like(logits)
        dloss = dloss.to(torch.float32).contiguous()
        _C.check(_C.lib().cx_xent_bwd(dloss.data_ptr(), logits.data_ptr(), int(logits.dtype == torch.bfloat16),
                                      lse.data_ptr(), labels.data_ptr(), dlogits.data_ptr(), N, V, logits.stride(0),
                                      dlogits.stride(0), scale, ignore_index, _C.cur_stream()), "cx_xent_bwd")
        return dlogits, None, None, None, None


class CrossEntropyLoss(torch.nn.Module):
    def __init__(self, ignore_index=-100, reduction="mean", label_smoothing=0.0, logit_scale=1.0, lse_square_scale=0.0,
                 inplace_backward=False, process_group=None, return_z_loss=False):
        super().__init__()
        if reduction not in ("mean", "none", "sum"):
            raise NotImplementedError("Only support reduction = 'mean' or 'none' or 'sum'")
        if label_smoothing != 0.0 or lse_square_scale != 0.0 or return_z_loss or process_group is not None:
            raise NotImplementedError("label smoothing, z-loss and tensor-parallel cross-entropy are not built")
        self.ignore_index, self.reduction = ignore_index, reduction
        self.logit_scale, self.inplace_backward = logit_scale, inplace_backward

    def forward(self, input, target):
        loss, _ = _Xent.apply(input, target, self.logit_scale, self.ignore_index, self.inplace_backward)
        if self.reduction == "mean":
            return loss.sum() / (target != self.ignore_index).sum()
        if self.reduction == "sum":
            return loss.sum()
        return loss
