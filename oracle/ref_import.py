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


def load_vit():
    """-> the reference's sc/models/vit/vit.py module, importable on a CPU-only host: `flash_attn` names come from
    this repo's shim (import only), `torchvision.ops.StochasticDepth` (p = 0 here) and `megablocks` are stubbed, and
    the attention core the reference takes from the third-party flash-attn package is replaced by the exact softmax
    attention it implements (fp32).  Everything else that runs is the reference's own python."""
    load()
    import torch

    import contrastors_amd.flash_attn_api as fa

    fa.install()

    def synth(name, path):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = M.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules.setdefault(name, m)
        return sys.modules[name]

    synth("contrastors.models", REF_ROOT / "models")
    synth("contrastors.models.vit", REF_ROOT / "models" / "vit")

    class StochasticDepth(torch.nn.Module):
        def __init__(self, p, mode):
            super().__init__()
            assert p == 0
            self.p = p

        def forward(self, x):
            return x

    class _Absent:
        def __init__(self, *a, **k):
            raise RuntimeError("megablocks is not available")

    ops = stub("torchvision.ops", StochasticDepth=StochasticDepth)
    stub("torchvision", ops=ops)
    layers = stub("megablocks.layers", dmoe=types.SimpleNamespace(dMoE=_Absent, ParallelDroplessMLP=_Absent))
    stub("megablocks", layers=layers)
    stub("megablocks.layers.arguments", Arguments=_Absent)
    stub("megablocks.layers.dmoe", dMoE=_Absent, ParallelDroplessMLP=_Absent)
    vit = importlib.import_module("contrastors.models.vit.vit")
    ratt = importlib.import_module("contrastors.layers.attention")

    def exact_qkvpacked(qkv, dropout_p=0.0, softmax_scale=None, causal=False, return_attn_probs=False, **kw):
        assert dropout_p == 0.0 and not causal
        q, k, v = qkv.unbind(2)  # (B, S, H, D)
        sc = float(softmax_scale) if softmax_scale is not None else q.shape[-1] ** -0.5
        att = torch.einsum("bshd,bthd->bhst", q, k) * sc
        return torch.einsum("bhst,bthd->bshd", att.softmax(-1), v)

    ratt.flash_attn_qkvpacked_func = exact_qkvpacked
    return vit


def load_biencoder():
    """-> the reference's sc/models/biencoder/modeling_biencoder.py (LogitScale, the pooling selectors incl.
    MultiHeadAttentionPooling, BiEncoder) importable on a CPU-only host: everything load_vit() arranges, plus import-only
    stand-ins for the MoE pieces of the text encoder package (megablocks, sc/layers/moe.py: out of scope, never called) and
    the exact softmax attention in place of the third-party kv-packed attention core the pooling head calls."""
    vit = load_vit()
    import torch

    for name, path in (("contrastors.models.biencoder", REF_ROOT / "models" / "biencoder"),):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m
    mb_layers = sys.modules["megablocks.layers"]
    for attr in ("moe", "mlp", "router", "common", "mpu"):
        if not hasattr(mb_layers, attr):
            setattr(mb_layers, attr, types.SimpleNamespace())
    if "contrastors.layers.moe" not in sys.modules:
        m = types.ModuleType("contrastors.layers.moe")
        m.__spec__ = M.ModuleSpec("contrastors.layers.moe", None)
        m.MoEBlock = object
        sys.modules["contrastors.layers.moe"] = m
    # `contrastors.models.vit` is the synthetic package of load_vit() (its __init__ wants timm): give it the names the
    # biencoder module imports from it -- the real ViTModel, and import-only placeholders for the hub config converters
    pkg = sys.modules["contrastors.models.vit"]
    pkg.ViTModel = vit.ViTModel
    for conv in ("clip_config_to_vit_config", "dino_config_to_vit_config", "hf_vit_config_to_vit_config", "timm_name_to_vit_config"):
        if not hasattr(pkg, conv):
            setattr(pkg, conv, None)
    bi = importlib.import_module("contrastors.models.biencoder.modeling_biencoder")
    ratt = importlib.import_module("contrastors.layers.attention")

    def exact_kvpacked(q, kv, dropout_p=0.0, softmax_scale=None, causal=False, **kw):
        assert dropout_p == 0.0 and not causal
        k, v = kv.unbind(2)  # (B, Sk, H, D)
        sc = float(softmax_scale) if softmax_scale is not None else q.shape[-1] ** -0.5
        att = torch.einsum("bshd,bthd->bhst", q, k) * sc
        return torch.einsum("bhst,bthd->bshd", att.softmax(-1), v)

    ratt.flash_attn_kvpacked_func = exact_kvpacked
    return bi


def load_bert_remap():
    """-> the reference's sc/models/encoder/bert.py (config conversion + HF <-> flash state-dict remapping), loaded by
    file path so that the package __init__ (which wants flash_attn) never runs."""
    load()
    for name, path in (("contrastors.models", REF_ROOT / "models"),
                       ("contrastors.models.encoder", REF_ROOT / "models" / "encoder")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [str(path)]
            sys.modules[name] = m
    return importlib.import_module("contrastors.models.encoder.bert")


def load_text_loader():
    """-> the reference's sc/dataset/text_text_loader.py.  `webdataset` (absent here) is replaced by the two helpers
    the loader takes from it, restated from their documented behaviour: `shardlists.expand_urls` = brace expansion,
    `tariterators.base_plus_ext` = split at the first dot of the last path component."""
    import re

    load()
    if "webdataset" not in sys.modules:
        def expand_urls(url):
            m = re.search(r"\{(\d+)\.\.(\d+)\}", url)
            if not m:
                return [url]
            a, b = m.group(1), m.group(2)
            return [url[: m.start()] + str(i).zfill(len(a)) + url[m.end():] for i in range(int(a), int(b) + 1)]

        def base_plus_ext(path):
            mm = re.match(r"^((?:.*/|)[^.]+)[.]([^/]*)$", path)
            return (mm.group(1), mm.group(2)) if mm else (None, None)

        w = types.ModuleType("webdataset")
        w.__spec__ = M.ModuleSpec("webdataset", None)
        w.shardlists = types.ModuleType("webdataset.shardlists")
        w.shardlists.expand_urls = expand_urls
        t = types.ModuleType("webdataset.tariterators")
        t.base_plus_ext = base_plus_ext
        w.tariterators = t
        sys.modules["webdataset"], sys.modules["webdataset.shardlists"], sys.modules["webdataset.tariterators"] = w, w.shardlists, t
    import fsspec
    from fsspec.implementations.local import LocalFileSystem

    class _NoS3(LocalFileSystem):  # the loader instantiates an "s3" filesystem before it looks at the spec (:214-215)
        def __init__(self, *a, config_kwargs=None, **k):
            super().__init__(*a, **k)

        @classmethod
        def _strip_protocol(cls, path):  # "s3://tmp/x" -> "/tmp/x": lets LocalShardDataset "download" local shards
            if isinstance(path, str) and path.startswith("s3://"):
                path = "/" + path[5:]
            return super()._strip_protocol(path)

    fsspec.register_implementation("s3", _NoS3, clobber=True)
    if "contrastors.dataset" not in sys.modules:
        pkg = types.ModuleType("contrastors.dataset")
        pkg.__path__ = [str(REF_ROOT / "dataset")]
        sys.modules["contrastors.dataset"] = pkg
    return importlib.import_module("contrastors.dataset.text_text_loader")
