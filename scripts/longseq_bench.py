"""cfg-3 shape (seq 2048) through the native trunk: one chunk forward + backward, time and share of the attention kernels.
usage: python scripts/longseq_bench.py [--seqs 16] [--seq-len 2048] [--checkpoint 0]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from contrastors_amd.nomic_bert import NomicBertConfig, NomicBertEngine, VarlenBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seqs", type=int, default=16)
ap.add_argument("--seq-len", type=int, default=2048)
ap.add_argument("--checkpoint", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
cfg = NomicBertConfig.nomic_bert_2048(rotary_emb_base=1000.0)
eng = NomicBertEngine(cfg, device="cuda", seed=0)
eng.train()
eng.gradient_checkpointing_enable(bool(a.checkpoint))
g = torch.Generator().manual_seed(0)
ids = torch.randint(1000, 30522, (a.seqs, a.seq_len), generator=g).cuda()
vb = VarlenBatch.from_lengths(ids, [a.seq_len] * a.seqs)
demb = torch.randn(a.seqs, cfg.n_embd, generator=g).cuda() * 1e-2


def step():
    emb, arena = eng.forward_chunk(vb, True)
    eng.backward_chunk(vb, arena, demb)


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
tok = a.seqs * a.seq_len
d, I, L, S = cfg.n_embd, cfg.n_inner, cfg.n_layer, a.seq_len
flop = 3 * L * tok * (2 * d * 3 * d + 2 * d * d + 2 * d * 2 * I + 2 * I * d) + (1 + 2.5) * L * a.seqs * 4.0 * S * S * d
flop *= (4.0 / 3.0) if a.checkpoint else 1.0
print(f"{a.seqs} x {S} tokens, 12 layers, checkpoint={a.checkpoint}: {ms:.1f} ms per forward+backward, {tok / ms:.0f} tokens/ms, "
      f"{flop / ms / 1e9:.0f} TFLOP/s")
