"""The `flash_attn` symbol surface (contrastors_amd.flash_attn_api) is sufficient for the REFERENCE's own layer
modules to import and construct (build container only: needs /root/reference; skipped on the GPU box).  Also checks the
pure-index helpers on CPU."""
import importlib
import importlib.machinery as M
import sys
import types
from pathlib import Path

import pytest
import torch

REF = Path("/root/reference/src/contrastors")


def test_bert_padding_roundtrip_cpu():
    from contrastors_amd.flash_attn_api.bert_padding import index_first_axis, pad_input, unpad_input

    mask = torch.tensor([[1, 1, 0], [1, 0, 0], [1, 1, 1]])
    h = torch.arange(18.0).view(3, 3, 2)
    u, idx, cu, mx = unpad_input(h, mask)  # 4-tuple, as the reference expects
    assert idx.tolist() == [0, 1, 3, 6, 7, 8] and cu.tolist() == [0, 2, 3, 6] and mx == 3
    assert cu.dtype == torch.int32
    back = pad_input(u, idx, 3, 3)
    assert torch.equal(back, h * mask.unsqueeze(-1))
    assert torch.equal(index_first_axis(h.view(9, 2), idx), u)


def test_surface_exports_every_symbol_contrastors_imports():
    import contrastors_amd.flash_attn_api as fa

    fa.install("flash_attn_cx_test")
    need = {
        "flash_attn_cx_test": ["flash_attn_qkvpacked_func", "flash_attn_varlen_qkvpacked_func",
                               "flash_attn_kvpacked_func", "flash_attn_varlen_kvpacked_func"],
        "flash_attn_cx_test.bert_padding": ["unpad_input", "pad_input", "index_first_axis"],
        "flash_attn_cx_test.ops.layer_norm": ["dropout_add_layer_norm", "dropout_add_layer_norm_parallel_residual",
                                              "layer_norm"],
        "flash_attn_cx_test.ops.rms_norm": ["RMSNorm", "rms_norm", "dropout_add_rms_norm",
                                            "dropout_add_rms_norm_parallel_residual"],
        "flash_attn_cx_test.ops.fused_dense": ["FusedDense"],
        "flash_attn_cx_test.ops.activations": ["swiglu"],
        "flash_attn_cx_test.layers.rotary": ["RotaryEmbedding", "apply_rotary_emb_func", "apply_rotary_emb_qkv_",
                                             "apply_rotary_emb_kv_"],
        "flash_attn_cx_test.losses.cross_entropy": ["CrossEntropyLoss"],
    }
    for mod, names in need.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), f"{mod}.{n}"


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_reference_layer_modules_import_and_construct_on_the_shim():
    import contrastors_amd.flash_attn_api as fa

    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("flash_attn", "contrastors", "torchvision",
                                                                          "wandb")}
    try:
        fa.install()

        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__spec__ = M.ModuleSpec(name, None)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        stub("wandb")
        tv = stub("torchvision")
        tv.ops = stub("torchvision.ops", StochasticDepth=torch.nn.Identity)
        pkg = types.ModuleType("contrastors")
        pkg.__path__ = [str(REF)]
        sys.modules["contrastors"] = pkg
        lay = types.ModuleType("contrastors.layers")
        lay.__path__ = [str(REF / "layers")]
        sys.modules["contrastors.layers"] = lay
        mlp = importlib.import_module("contrastors.layers.mlp")
        emb = importlib.import_module("contrastors.layers.embedding")
        importlib.import_module("contrastors.layers.attention")
        g = mlp.GatedMLP(256, hidden_features=512, bias1=False, bias2=False, activation=torch.nn.functional.silu,
                         fused_bias_fc=True)
        from contrastors_amd.flash_attn_api.ops.fused_dense import FusedDense

        assert isinstance(g.fc11, FusedDense) and g.fc11.weight.shape == (512, 256)
        rot = emb.VarLengthRotaryEmbedding(dim=64, base=1000.0, interleaved=False)
        rot._update_cos_sin_cache(128, device=torch.device("cpu"), dtype=torch.float32)
        assert rot._cos_cached.shape == (128, 32) and rot.inv_freq.shape == (32,)
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("flash_attn", "contrastors", "torchvision", "wandb")]:
            del sys.modules[k]
        sys.modules.update(saved)


class _CpuKernels:
    """Test-only stand-ins for the kernel layer UNDER the shim's python (the autograd Functions that call the C-ABI): plain
    torch on CPU.  Everything above them -- the public functions with the reference's argument conventions, FusedDense,
    RotaryEmbedding, bert_padding -- is the shipped code, called by the reference's own model python."""

    @staticmethod
    def varlen_qkv(qkv, cu_seqlens, max_seqlen, scale, dropout_p=0.0):
        assert qkv.dim() == 4 and qkv.shape[1] == 3 and cu_seqlens.dtype == torch.int32 and dropout_p == 0.0
        out = torch.empty(qkv.shape[0], qkv.shape[2], qkv.shape[3], dtype=qkv.dtype)
        for b in range(cu_seqlens.numel() - 1):
            s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
            assert e - s <= max_seqlen
            q, k, v = qkv[s:e, 0].float(), qkv[s:e, 1].float(), qkv[s:e, 2].float()
            p = torch.softmax(torch.einsum("qhd,khd->hqk", q, k) * scale, dim=-1)
            out[s:e] = torch.einsum("hqk,khd->qhd", p, v).to(qkv.dtype)
        return out

    @staticmethod
    def fused_dense(x, weight, bias):
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def ln(x0, residual, weight, bias, eps, prenorm, residual_in_fp32, rms=False):
        assert not rms
        z = x0.float() + (residual.float() if residual is not None else 0)
        out = torch.nn.functional.layer_norm(z, (z.shape[-1],), weight.float(), bias.float(), eps).to(x0.dtype)
        zdt = torch.float32 if (residual_in_fp32 or (residual is not None and residual.dtype == torch.float32)) else x0.dtype
        return (out, z.to(zdt)) if prenorm else out

    @staticmethod
    def swiglu(x, y):
        return torch.nn.functional.silu(x.float()).to(x.dtype) * y

    @staticmethod
    def rotary_varlen(x, cos, sin, cu_seqlens, max_seqlen, inplace):
        # x (T, H, D) or (B, S, H, D); non-interleaved; positions restart at every sequence of cu_seqlens
        ro = cos.shape[-1] * 2
        out = x if inplace else x.clone()
        if cu_seqlens is None:
            S = x.shape[1]
            c, s = cos[:S].float()[None, :, None, :], sin[:S].float()[None, :, None, :]
            x1, x2 = x[..., : ro // 2].float(), x[..., ro // 2: ro].float()
            out[..., :ro] = torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1).to(x.dtype)
            return out
        for b in range(cu_seqlens.numel() - 1):
            s0, e0 = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
            n = e0 - s0
            assert n <= max_seqlen
            c, s = cos[:n].float()[:, None, :], sin[:n].float()[:, None, :]
            x1, x2 = x[s0:e0, :, : ro // 2].float(), x[s0:e0, :, ro // 2: ro].float()
            out[s0:e0, :, :ro] = torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1).to(x.dtype)
        return out

    @staticmethod
    def rotary_qkv(qkv, cos, sin):
        # non-interleaved rotation of q and k by position (fixed-length (B,S,3,H,D) or the caller's layout)
        ro = cos.shape[-1] * 2
        x = qkv[..., :2, :, :ro].float()
        x1, x2 = x[..., : ro // 2], x[..., ro // 2:]
        S = qkv.shape[-4]
        c, s = cos[:S].float()[:, None, None, :], sin[:S].float()[:, None, None, :]
        qkv[..., :2, :, :ro] = torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1).to(qkv.dtype)
        return qkv


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_reference_nomic_bert_model_runs_on_the_shim_surface_and_matches_the_oracle(monkeypatch):
    """The reference's OWN flash model python (contrastors.models.encoder.modeling_nomic_bert.NomicBertModel: Block,
    FlashAttention, GatedMLP, BertEmbeddings, unpad / pad plumbing) imported on `flash_attn` = this package's shim, with
    the kernel layer replaced by torch-CPU stand-ins: every call the reference makes into the flash_attn surface (names,
    positional / keyword arguments, tuple arities, dtypes of cu_seqlens, shapes) goes through the shipped wrappers, and the
    hidden states must equal the oracle's.  The same composition runs on the real kernels in tests/test_shim_compose_gpu.py."""
    from types import SimpleNamespace

    from transformers import GPT2Config, PreTrainedModel  # noqa: F401  (before torchvision is stubbed: transformers probes it)

    import contrastors_amd.flash_attn_api as fa
    from contrastors_amd.flash_attn_api import flash_attn_interface as fi
    from contrastors_amd.flash_attn_api.layers import rotary as frot
    from contrastors_amd.flash_attn_api.ops import activations as fact
    from contrastors_amd.flash_attn_api.ops import fused_dense as fd
    from contrastors_amd.flash_attn_api.ops import layer_norm as fln
    from oracle import encoder_ref
    from oracle.make_golden import TINY_NOMIC

    monkeypatch.setattr(fi._VarlenQKVPacked, "apply", _CpuKernels.varlen_qkv)
    monkeypatch.setattr(fd._FusedDenseFn, "apply", _CpuKernels.fused_dense)
    monkeypatch.setattr(fln._DropoutAddLN, "apply", _CpuKernels.ln)
    monkeypatch.setattr(fact._SwiGLU, "apply", _CpuKernels.swiglu)
    monkeypatch.setattr(frot._ApplyRotaryQKV, "apply", _CpuKernels.rotary_qkv)
    monkeypatch.setattr(frot._ApplyRotary, "apply", _CpuKernels.rotary_varlen)

    prefixes = ("flash_attn", "contrastors", "torchvision", "wandb", "megablocks")
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in prefixes}
    try:
        fa.install()

        def stub(name, **attrs):
            m = types.ModuleType(name)
            m.__spec__ = M.ModuleSpec(name, None)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        stub("wandb")
        tv = stub("torchvision")
        tv.ops = stub("torchvision.ops", StochasticDepth=torch.nn.Identity)
        mb = stub("megablocks")   # MoE: out of scope, only its import must succeed
        mb.layers = stub("megablocks.layers", dmoe=types.SimpleNamespace(), moe=types.SimpleNamespace())
        stub("megablocks.layers.arguments", Arguments=object)
        for name in ("contrastors", "contrastors.layers", "contrastors.models", "contrastors.models.encoder"):
            pkg = types.ModuleType(name)
            pkg.__path__ = [str(REF.joinpath(*name.split(".")[1:]))]
            sys.modules[name] = pkg
        stub("contrastors.layers.moe", MoEBlock=object)
        blk = importlib.import_module("contrastors.layers.block")
        sys.modules["contrastors.layers"].Block = blk.Block
        modeling = importlib.import_module("contrastors.models.encoder.modeling_nomic_bert")
        cfg_mod = importlib.import_module("contrastors.models.encoder.configuration_nomic_bert")

        ns = SimpleNamespace(**TINY_NOMIC)
        cfg = cfg_mod.NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items()})
        torch.manual_seed(0)
        model = modeling.NomicBertModel(cfg, add_pooling_layer=False).eval()
        sd = encoder_ref.random_state_dict(ns, 5)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all("rotary" in k or "inv_freq" in k for k in missing), missing

        g = torch.Generator().manual_seed(1)
        B, S = 5, 24
        lens = torch.tensor([24, 3, 17, 24, 9])
        ids = torch.randint(5, ns.vocab_size, (B, S), generator=g)
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        ids = ids * mask
        with torch.no_grad():
            out = model(ids, attention_mask=mask)
            hidden = out.last_hidden_state if hasattr(out, "last_hidden_state") else out[0]
            want = encoder_ref.encoder_hidden_states(sd, ns, ids, mask)
        keep = mask.bool()
        err = float((hidden.float() - want)[keep].abs().max())
        assert err < 2e-4, err
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in prefixes]:
            del sys.modules[k]
        sys.modules.update(saved)
