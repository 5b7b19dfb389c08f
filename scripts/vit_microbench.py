"""ViT-B/16 image tower (cx_vit_forward / cx_vit_backward): images/s for forward-only (LiT's frozen tower) and
forward+backward (CLIP).  usage: python scripts/vit_microbench.py [--batch 512]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd.vit import ViTConfig, ViTEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
cfg = ViTConfig.vit_base_patch16_224()
eng = ViTEngine(cfg, device="cuda", seed=0).train()
pix = torch.randn(a.batch, 3, 224, 224, device="cuda").to(torch.bfloat16)
probe = torch.randn(a.batch, cfg.n_embd, device="cuda")
# forward FLOPs per image: 12 layers x (GEMM params 7.08 M x 2 + attention 4*S*d) x 197 tokens + patch projection
S, d = 197, 768
per_tok = 12 * (2 * (d * 3 * d + d * d + 2 * d * 3072) + 4 * S * d)
fwd_flop = a.batch * (S * per_tok + 196 * 2 * 768 * 768)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


def fwd_only():
    eng.forward_chunk(pix, False)


def fwd_bwd():
    emb, arena = eng.forward_chunk(pix, True)
    eng.backward_chunk(a.batch, arena, probe)


t_f, t_fb = timed(fwd_only), timed(fwd_bwd)
print(f"ViT-B/16 224x224, batch {a.batch} ({a.batch * S} tokens)")
print(f"forward only : {t_f:8.2f} ms  {a.batch / t_f * 1e3:9.0f} img/s  {fwd_flop / t_f / 1e9:7.1f} TFLOP/s")
print(f"forward+backward: {t_fb:8.2f} ms  {a.batch / t_fb * 1e3:9.0f} img/s  {3 * fwd_flop / t_fb / 1e9:7.1f} TFLOP/s")
