"""BASELINE configs[4] (CLIP-style ViT-B/16 + text tower, global batch 32768 = 4096 images per GPU) at its per-GPU SHAPE:
the image tower's forward + backward in ONE direct step (the reference refuses GradCache for image-text,
sc/trainers/image_text.py:154-157, and relies on activation checkpointing, sc/models/vit/vit.py:200-231) and the
4096 x 32768 symmetric loss on the fp8 similarity path.

Size-independent parity: the batch is 8 distinct images repeated 512 times, so every embedding must equal its b = 8
counterpart and the weight gradient must be 512 x the b = 8 gradient (linearity of the backward in the batch)."""
import gc

import pytest
import torch

from tests.gpu_util import rel_err, report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_vit_b16_4096_images_checkpointed_fits_and_is_linear_in_the_batch():
    from contrastors_amd.vit import ViTConfig, ViTEngine

    cfg = ViTConfig.vit_base_patch16_224()
    eng = ViTEngine(cfg, device=DEV, pooling="cls", seed=3)
    eng.train()
    eng.gradient_checkpointing_enable(True)
    g = torch.Generator().manual_seed(21)
    base = torch.randn(8, 3, 224, 224, generator=g).to(DEV).bfloat16()
    probe8 = torch.randn(8, cfg.n_embd, generator=g).to(DEV)

    emb8, arena = eng.forward_chunk(base, True)
    emb8 = emb8.clone()
    eng.zero_grad()
    eng.backward_chunk(base, arena, probe8)
    g8 = eng.flat_grad.clone()
    del arena
    eng._arena_free.clear()
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()

    R = 512
    pixels = base.repeat(R, 1, 1, 1)          # 4096 x 3 x 224 x 224 bf16 = 1.2 GB, 806 912 tokens
    probe = probe8.repeat(R, 1)
    emb, arena = eng.forward_chunk(pixels, True)
    eng.zero_grad()
    eng.backward_chunk(pixels, arena, probe)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated()
    report("cfg5_vit_b4096", peak_gb=peak / 1e9, tower_delta_gb=(peak - before) / 1e9, tokens=4096 * 197)
    assert peak < 200e9, "one tower of the CLIP step must leave room for the other tower and the loss on 288 GB"

    assert torch.isfinite(emb).all()
    e = float((emb.view(R, 8, -1) - emb8[None]).abs().max())
    assert e < 2e-2, e          # same kernels, different tile / split-K partitions at M = 806 912
    gbig = eng.flat_grad
    assert torch.isfinite(gbig).all()
    err = rel_err(gbig / R, g8)
    report("cfg5_vit_b4096_grad", rel_err_vs_8_images_times_512=err)
    assert err < 2e-2, err
    eng._arena_free.clear()
    del eng, arena, pixels, emb, gbig
    gc.collect()
    torch.cuda.empty_cache()


def test_fp8_symmetric_loss_at_cfg5_shape():
    """4096 image embeddings against 32768 gathered text embeddings (and the transpose direction the DualEncoder adds):
    the fp8 path against the exact path at the shape's size, tolerance as in tests/test_infonce_fp8_gpu.py."""
    from contrastors_amd.loss import clip_loss

    g = torch.Generator().manual_seed(4)
    N, G, d = 4096, 32768, 768
    docs = torch.nn.functional.normalize(torch.randn(G, d, generator=g), dim=-1)
    q = torch.nn.functional.normalize(0.15 * docs[:: G // N] + torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=-1), dim=-1)
    out = {}
    for fp8 in (False, True):
        qq = q.to(DEV).requires_grad_()
        dd = docs.to(DEV).requires_grad_()
        loss = clip_loss(qq, dd, 50.0, use_fp8=fp8)
        loss.backward()
        out[fp8] = (loss.item(), qq.grad.clone(), dd.grad.clone())
    l0, l1 = out[False][0], out[True][0]
    eq, ed = rel_err(out[True][1], out[False][1]), rel_err(out[True][2], out[False][2])
    # an INDEPENDENT reference at the shape's own size (the fp8 path had only been held against fp64 up to
    # 512 x 2048): plain torch in fp64 on the device -- the (4096, 32768) logit matrix is 1 GB there
    qr, dr = q.to(DEV).double().requires_grad_(), docs.to(DEV).double().requires_grad_()
    ref = torch.nn.functional.cross_entropy(qr @ dr.T * 50.0, torch.arange(N, device=DEV) * (G // N))
    ref.backward()
    e64 = {fp8: (abs(out[fp8][0] - ref.item()), rel_err(out[fp8][1], qr.grad), rel_err(out[fp8][2], dr.grad)) for fp8 in out}
    report("cfg5_fp8_loss", exact=l0, fp8=l1, fp64=ref.item(), rel_dq=eq, rel_dd=ed, exact_vs_fp64=list(e64[False]),
           fp8_vs_fp64=list(e64[True]))
    assert abs(l1 - l0) < 2e-2
    assert eq < 8e-2 and ed < 8e-2
    assert e64[False][0] < 1e-5 * abs(ref.item()) + 1e-6 and e64[False][1] < 1e-4 and e64[False][2] < 1e-4, e64[False]
    assert e64[True][0] < 2e-2 and e64[True][1] < 8e-2 and e64[True][2] < 8e-2, e64[True]
