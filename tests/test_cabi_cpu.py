"""CPU-side checks of the boundary: the shared object loads, exports every symbol include/contrastors_hip.h declares,
and the ctypes struct mirrors have the C compiler's layout.  No compute call is made (no GPU here)."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HDR = ROOT / "include" / "contrastors_hip.h"


@pytest.fixture(scope="module")
def built():
    from contrastors_amd import build

    return build.build()


def test_header_symbols_exported(built):
    from contrastors_amd import _C

    text = HDR.read_text()
    declared = set(re.findall(r"\b(cx_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    lib = _C.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_C.EXPORTED_SYMBOLS), declared ^ set(_C.EXPORTED_SYMBOLS)
    assert lib.cx_abi_version() == 10  # 10: cx_gemm_bf16_act_bwd; 9: cx_infonce_fwd_argmax; 8: batched cast-transpose; 7: PatchDropout fields; 2: CxChunkBuffers.checkpoint; 3: dropout state + sorted embedding backward; 4: attn_pdrop; 5: layer_events; 6: ckpt_keep
    assert b"gfx950" in lib.cx_build_info()
    assert lib.cx_error_string(-1) == b"unsupported shape"
    assert lib.cx_infonce_ws_floats(2048, 16384) == 2048 * (2 * 2 * 128 + 1)
    # the product library carries no debug switch, probe or superseded kernel generation
    for name in _C.DEV_EXPORTED_SYMBOLS:
        assert not hasattr(lib, name), f"{name} must live in the dev library only"


def test_dev_header_symbols_exported(built):
    from contrastors_amd import _C

    text = (ROOT / "include" / "contrastors_hip_dev.h").read_text()
    declared = set(re.findall(r"\b(cx_[a-z0-9_]+)\s*\(", text))
    assert declared == set(_C.DEV_EXPORTED_SYMBOLS), declared ^ set(_C.DEV_EXPORTED_SYMBOLS)
    dev = _C.dev_lib()
    for name in sorted(declared | set(_C.EXPORTED_SYMBOLS)):
        assert hasattr(dev, name), f"{name} missing from the dev library"


def test_ctypes_struct_layout_matches_c(tmp_path, built):
    from contrastors_amd import _C

    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "contrastors_hip.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(CxLayerWeights), sizeof(CxEncoderDesc),"
        " sizeof(CxChunkBuffers), offsetof(CxEncoderDesc, layers), offsetof(CxEncoderDesc, word_emb),"
        " offsetof(CxChunkBuffers, delta), offsetof(CxEncoderDesc, patch_dim), offsetof(CxEncoderDesc, Wpatch),"
        " offsetof(CxChunkBuffers, patch_proj), offsetof(CxChunkBuffers, ckpt_keep)); printf(\"%zu %zu\\n\", offsetof(CxChunkBuffers, patch_keep), offsetof(CxChunkBuffers, n_patch_all)); return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    want = [C.sizeof(_C.CxLayerWeights), C.sizeof(_C.CxEncoderDesc), C.sizeof(_C.CxChunkBuffers),
            _C.CxEncoderDesc.layers.offset, _C.CxEncoderDesc.word_emb.offset, _C.CxChunkBuffers.delta.offset,
            _C.CxEncoderDesc.patch_dim.offset, _C.CxEncoderDesc.Wpatch.offset, _C.CxChunkBuffers.patch_proj.offset,
            _C.CxChunkBuffers.ckpt_keep.offset, _C.CxChunkBuffers.patch_keep.offset, _C.CxChunkBuffers.n_patch_all.offset]
    assert got == want


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from contrastors_amd import _C

    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(ImportError):
        _C.lib()


def test_engine_refuses_cpu():
    from contrastors_amd.nomic_bert import NomicBertConfig, NomicBertEngine

    with pytest.raises(RuntimeError):
        NomicBertEngine(NomicBertConfig(n_layer=1), device="cpu")


def test_product_never_imports_oracle():
    for p in (ROOT / "contrastors_amd").rglob("*.py"):
        assert "oracle" not in re.sub(r'""".*?"""', "", p.read_text(), flags=re.S).replace("# oracle", ""), p


def test_v6_accumulator_map_is_a_bijection_and_the_generated_file_is_current():
    """scripts/gen_v6_acc.py: the 8 x 8 blocks of 16 x 16 of a wave's sub-tile cover AGPRs 0 .. 255 exactly once in quads, region
    i = 4b + a (32 x 32) owns a[16i : 16i + 15] with piece q = 2 (mb & 1) + (nb & 1) in its q-th quad, and the committed
    gemm_v6_acc.inc is what the generator writes."""
    import importlib.util
    import re
    root = Path(__file__).resolve().parent.parent
    inc = root / "contrastors_amd" / "csrc" / "gemm_v6_acc.inc"
    before = inc.read_text()
    spec = importlib.util.spec_from_file_location("gen_v6_acc", root / "scripts" / "gen_v6_acc.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # (re-writes the file: must be a no-op)
    assert inc.read_text() == before, "gemm_v6_acc.inc is stale: run scripts/gen_v6_acc.py"
    seen = set()
    for mb in range(8):
        for nb in range(8):
            lo = mod.base(mb, nb)
            assert lo % 4 == 0 and lo // 16 == 4 * (mb >> 1) + (nb >> 1) and (lo % 16) // 4 == 2 * (mb & 1) + (nb & 1)
            seen.update(range(lo, lo + 4))
    assert seen == set(range(256))
    cases = re.findall(r"case (\d+): asm volatile\(\"v_mfma_f32_16x16x32_bf16 a\[(\d+):(\d+)\], %0, %1, a\[(\d+):(\d+)\]\"", before)
    assert len(cases) == 64 and all(int(lo) == mod.base(int(i) >> 3, int(i) & 7) == int(clo) and int(hi) == int(lo) + 3 == int(chi)
                                    for i, lo, hi, clo, chi in cases)
