"""flash_attn.bert_padding: K4 (sc/models/encoder/modeling_nomic_bert.py:332-333,392-393).  Index plumbing in torch;
`unpad_input` returns the 4-tuple the reference relies on."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def index_first_axis(x, indices):
    return x.reshape(x.shape[0], -1).index_select(0, indices.long()).reshape(-1, *x.shape[1:])


def unpad_input(hidden_states, attention_mask):
    seqlens = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(attention_mask.flatten(), as_tuple=False).flatten()
    max_seqlen = int(seqlens.max().item())
    cu_seqlens = F.pad(torch.cumsum(seqlens, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(-1, *hidden_states.shape[2:])
    return flat.index_select(0, indices), indices, cu_seqlens, max_seqlen


def pad_input(hidden_states, indices, batch, seqlen):
    out = torch.zeros(batch * seqlen, *hidden_states.shape[1:], dtype=hidden_states.dtype, device=hidden_states.device)
    out = out.index_copy(0, indices.long(), hidden_states)
    return out.view(batch, seqlen, *hidden_states.shape[1:])
