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
