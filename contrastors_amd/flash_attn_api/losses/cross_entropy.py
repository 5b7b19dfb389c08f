"""flash_attn.losses.cross_entropy.CrossEntropyLoss: K12, MLM only (sc/models/encoder/modeling_nomic_bert.py:47,606).
Outside the contrastive hot path: the symbol exists so the reference module imports; it is plain
torch cross-entropy, not a HIP kernel, and is NOT part of any measured path."""
import torch


class CrossEntropyLoss(torch.nn.CrossEntropyLoss):
    def __init__(self, ignore_index=-100, reduction="mean", label_smoothing=0.0, inplace_backward=False, **kw):
        super().__init__(ignore_index=ignore_index, reduction=reduction, label_smoothing=label_smoothing)
