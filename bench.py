"""bench.py -- query-doc pairs/sec of the contrastive GradCache training step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torch.distributed environment re-launches ITSELF as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
(one rank per GPU over RCCL); launched by the driver under torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.  Either way rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): nomic-bert-2048 architecture (random init, no hub), bi-encoder with mean pooling,
paired InfoNCE at logit scale 50, seq_len 128, GLOBAL batch 16384 (fixed; each of the N ranks owns 16384/N pairs ->
"strong" scaling), GradCache chunked re-forward, bf16 MFMA compute with fp32 master weights, grad-clip 1.0, AdamW,
cosine schedule.  A step = everything in the reference's training_step (sc/trainers/base.py:366-393): GradCache pass 1,
embedding all-gather, fused loss fwd/bwd, pass 2 (re-forward + backward), gradient all-reduce, clip, AdamW, scheduler,
bf16 shadow refresh.  Synthetic token ids are staged in HBM before the timed region.

The JSON line carries, besides the contract fields:
  roofline      dominant kernel family (bf16 MFMA GEMMs) timed live with HIP events on its own stream (every 7th launch)
  weak          the same step at 2048 pairs per GPU (global batch 2048 x N: SURVEY.md §8(d) asks for both curves)
  resident      2048 pairs per GPU with pass 1's activations kept in HBM (no re-forward; identical results)
  dropin_chunk64  (N = 1) the same step at the reference recipe's GradCache chunk_size 64 (contrastive_pretrain.yaml:15)
  exchange      (N > 1) what carries the embedding exchange: set up, verified bit-exact against RCCL, timed, faster one chosen
  xgmi_allgather  (N > 1) the embedding all-gather of the loss timed on its own, against 7 x 153 GB/s of xGMI per GPU
  cfg1 / cfg3 / lit / clip  (N = 1) the other BASELINE configs at their per-GPU shapes, each with its own roofline fraction
  box           what THIS box sustains, measured right before the metric: register-only bf16 MFMA probe (TFLOP/s), HBM copy
                (TB/s), and socket power / shader clock sampled during the timed region; roofline.frac_of_box_ceiling =
                achieved / box.mfma_probe_tflops (flat keys box_*, frac_of_box_ceiling)
  schedule_per_rank / allreduce_exposed_ms / allgather_us_* / allgather_xgmi_frac_*  (N > 1) one line answers >= 70 % xGMI, >= 6 x
  cpu_baseline  (N = 1) oracle = CPU restatement of the reference: one 64-pair cfg-2 chunk forward + backward, and the
                cfg-1 full step (bert-base, B = 32, S = 64: forward, backward, InfoNCE, AdamW) on the host cores
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# the host driver only supports dmabuf IPC: must be in the environment before the HIP runtime starts (RCCL, N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0      # MI355X dense fp8 MFMA
GFLOP_PER_PAIR = 236.84       # 2 seqs x 4 fwd-equivalents x 29.595 GFLOP + loss
XGMI_PEAK_GBS = 7 * 153.0     # per GPU, each direction
WEAK_PAIRS_PER_GPU = 2048     # per-GPU b = 2048 fixed


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--global-batch", type=int, default=16384)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--chunk-size", type=int, default=int(os.environ.get("CX_BENCH_CHUNK", 4096)),
                    help="GradCache chunk = sequences per encoder call.  A pure memory knob (results are identical); the "
                         "reference recipe uses 64 on 80 GB parts (contrastive_pretrain.yaml:15); 4096 = 524288 token rows per "
                         "GEMM launch (2048 whole 256-row tile panels), peaks at ~175 GB of the MI355X's 288 GB and measures +0.5 ... "
                         "+1.0 % over 2048 on the same box (round 6: half the launch boundaries and tile-round tails; rounds 2-5 ran 2048)")
    ap.add_argument("--layers", type=int, default=12, help=argparse.SUPPRESS)  # debugging only; 12 = the metric
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the per-box MFMA / HBM probes and the power / clock sampler")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the weak-scaling and chunk-64 records")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the cfg1 / cfg3 / lit / clip records (N = 1)")
    ap.add_argument("--only-config-legs", type=str, default="", help=argparse.SUPPRESS)  # profiling: e.g. cfg3 or lit,clip
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    # every 7th GEMM launch is timed with HIP events: a stride coprime to the launch pattern's periods (4 GEMMs per layer
    # forward, 8 per layer backward) so that the sample walks through every shape; a stride of 8 always lands on the
    # same two (Wqkv forward, fc2-dgrad + SwiGLU backward) and over-reads the family average by ~4 %
    ap.add_argument("--prof-stride", type=int, default=7)
    return ap.parse_args()


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (a 256-thread OpenMP pool on
    an 8-core quota would spin for minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline_subprocess(seq_len: int, timeout_s: int = 300) -> dict:
    """Run the CPU leg in a child process with a hard time bound so the GPU line is never lost to a slow host."""
    try:
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-only", "--seq-len",
                            str(seq_len)], capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu leg failed: {r.stderr.strip()[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "pairs/s", "cores": usable_cores(), "kind": "port",
                "sample": f"cpu leg exceeded {timeout_s} s"}


def cpu_baseline(seq_len: int) -> dict:
    """Reference algorithm (oracle restatement, fp32, torch CPU kernels) on a bounded sample of the same workload:
    ONE GradCache chunk of the reference recipe -- 64 query-document pairs, direct forward + backward through the
    12-layer encoder + InfoNCE -- min of 3 repetitions after an 8-pair warm-up.  A port, not the
    reference itself: /root/reference does not exist on the GPU box (the oracle is pinned to it by tests/golden)."""
    import torch

    from contrastors_amd.nomic_bert import NomicBertConfig
    from oracle import encoder_ref, infonce_ref

    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = NomicBertConfig.nomic_bert_2048()
    ns = SimpleNamespace(**{k: getattr(cfg, k) for k in cfg.__dataclass_fields__})
    sd = encoder_ref.random_state_dict(ns, 0)
    for v in sd.values():
        v.requires_grad_(True)
    pairs = 64
    g = torch.Generator().manual_seed(1234)
    q = torch.randint(1000, 30522, (pairs, seq_len), generator=g)
    d = torch.randint(1000, 30522, (pairs, seq_len), generator=g)
    mask = torch.ones(pairs, seq_len, dtype=torch.long)

    def step(n):
        loss = infonce_ref.clip_loss_ref(encoder_ref.biencoder_embedding(sd, ns, q[:n], mask[:n]),
                                         encoder_ref.biencoder_embedding(sd, ns, d[:n], mask[:n]), 50.0)
        loss.backward()

    step(8)  # warm-up (allocator, thread pool)
    reps, times = 3, []
    for _ in range(reps):
        t0 = time.perf_counter()
        step(pairs)
        times.append(time.perf_counter() - t0)
    dt = min(times)
    out = {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": f"one {pairs}-pair chunk x seq {seq_len}, direct fwd+bwd+InfoNCE (3x fwd FLOPs, no GradCache "
                     f"re-forward), fp32 torch-CPU oracle port (the reference tree is absent on the GPU box), min of "
                     f"{reps} reps = {dt:.2f} s (all: {', '.join(f'{t:.2f}' for t in times)})"}
    # BASELINE configs[0] / SURVEY.md §8(d): the reference's own CPU-runnable case as a FULL step -- bert-base-uncased
    # bi-encoder, paired InfoNCE, batch 32, seq 64: forward, backward, loss, clip, AdamW on the host cores
    del sd
    cfg1 = NomicBertConfig.bert_base_uncased()
    ns1 = SimpleNamespace(**{k: getattr(cfg1, k) for k in cfg1.__dataclass_fields__})
    sd1 = encoder_ref.random_state_dict(ns1, 0)
    params = [v.requires_grad_(True) for v in sd1.values()]
    opt = torch.optim.AdamW(params, lr=2e-5, weight_decay=0.01)
    B1, S1 = 32, 64
    q1 = torch.randint(1000, 30522, (B1, S1), generator=g)
    d1 = torch.randint(1000, 30522, (B1, S1), generator=g)
    m1 = torch.ones(B1, S1, dtype=torch.long)

    def step1():
        opt.zero_grad(set_to_none=True)
        loss = infonce_ref.clip_loss_ref(encoder_ref.biencoder_embedding(sd1, ns1, q1, m1),
                                         encoder_ref.biencoder_embedding(sd1, ns1, d1, m1), 50.0)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()

    step1()
    t1 = []
    for _ in range(3):
        t0 = time.perf_counter()
        step1()
        t1.append(time.perf_counter() - t0)
    out["cfg1_full_step"] = {"value": B1 / min(t1), "unit": "pairs/s", "ms_per_step": 1e3 * min(t1), "cores": cores,
                             "sample": f"configs[0]: bert-base bi-encoder, B = {B1}, S = {S1}, full step (fwd + bwd + InfoNCE + clip + "
                                       f"AdamW), fp32 torch-CPU oracle port, min of 3 = {min(t1):.2f} s"}
    return out


# ---------------------------------------------------------------------------------------- the other BASELINE configs
# Each leg runs its config's per-GPU SHAPE on one GPU (synthetic inputs, random-init weights of the named architecture) as a
# full training step and reports examples/s + the fraction of the bf16 MFMA peak its ALGORITHMIC encoder FLOPs amount to
# (re-computation under activation checkpointing is not counted as work).
def _fwd_flop_per_token(d, mlp_cols, S):   # 2 * (Wqkv + Wout + fc1 + fc2) + QK^T and PV
    return 2.0 * d * (3 * d + d + mlp_cols) + 4.0 * S * d


def _time_steps(torch, step, warmup, steps):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))


def _pct(sorted_ms, q):
    return sorted_ms[min(len(sorted_ms) - 1, max(0, int(round(q * (len(sorted_ms) - 1)))))]


def leg_cfg1(torch, dev, steps, hf_dropout=True):
    """configs[0] on the HIP path (its CPU twin is cpu_baseline.cfg1_full_step): bert-base bi-encoder, B = 32, S = 64, direct
    step.  M = 2048 token rows per GEMM launch: the launch-bound end of the design, reported so that it is on record.
    hf_dropout = True (the record): the tower the reference trains -- its conversion of the hub config carries
    hidden_dropout_prob = attention_probs_dropout_prob = 0.1 (sc/models/encoder/bert.py:19-21); False: the architecture alone."""
    from contrastors_amd.config import Config, DataArgs, ModelArgs, TrainArgs
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.trainers import TextTextTrainer

    cfg = Config(train_args=TrainArgs(learning_rate=2e-5, weight_decay=0.01, warmup_steps=1, grad_cache=False,
                                      schedule_type="linear", max_grad_norm=1.0, clamp_logits=False),
                 data_args=DataArgs(batch_size=32, seed=3),
                 model_args=ModelArgs(logit_scale=50.0, pooling="mean", model_name="bert-base-uncased", seq_len=64))
    tr = TextTextTrainer(cfg, torch.bfloat16, device=dev, trunk_config=NomicBertConfig.bert_base_uncased(hf_dropout=hf_dropout),
                         total_steps=1000)
    g = torch.Generator().manual_seed(7)
    B, S = 32, 64
    batch = {}
    for side in ("query", "document"):
        batch[f"{side}_input_ids"] = torch.randint(1000, 30522, (B, S), generator=g).to(dev)
        batch[f"{side}_seqlens"] = [S] * B
    ms = _time_steps(torch, lambda: tr.training_step(batch), 3, max(steps, 10))
    flop = 3 * 2 * B * S * 12 * _fwd_flop_per_token(768, 2 * 3072, S)
    med = _pct(ms, 0.5)
    return {"workload": "configs[0]: bert-base-uncased bi-encoder, paired InfoNCE, B = 32, S = 64, direct step (HIP path)",
            "value": B / (med * 1e-3), "unit": "pairs/s", "ms_per_step": med, "p10_ms": _pct(ms, 0.1), "p90_ms": _pct(ms, 0.9),
            "steps": len(ms), "frac_of_mfma_peak": flop / (med * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
            "dropout": 0.1 if hf_dropout else 0.0,
            "note": "2048 token rows per GEMM launch: far below one round of 256 x 256 tiles on 256 CUs; the tower is "
                    + ("bert-base-uncased as the reference converts its hub config: residual, embedding and attention dropout 0.1 "
                       "(sc/models/encoder/bert.py:19-21), Philox masks regenerated in backward"
                       if hf_dropout else "the BERT-base ARCHITECTURE with dropout 0")}


def _selective(rec):
    """What the `selective_checkpointing` sub-record of a leg keeps of a second run of the same leg with
    train_args.checkpoint_keep_layers = "auto" (the library default; the leg's own record takes the recipe literally)."""
    keys = ("value", "unit", "ms_per_step", "p10_ms", "p90_ms", "steps", "peak_hbm_gb", "kept_blocks")
    out = {k: rec[k] for k in keys if k in rec}
    out["frac_of_mfma_peak"] = rec["roofline"]["frac"]
    out["note"] = ("same step, same results bit for bit: the warm-up step takes the recipe literally and measures the step's peak "
                   "HBM, from then on the top blocks keep their activations as far as 90 % of the device minus that peak "
                   "allows (CxChunkBuffers.ckpt_keep) and only the blocks below them are recomputed in backward")
    return out


def leg_cfg3(torch, dev, steps, keep=0):
    """configs[2] per-GPU shape: 32 queries + 256 documents (1 positive + 7 hard negatives each) x 2048 tokens, Matryoshka
    {768, 512, 256, 128}, hamming, activation checkpointing, direct step (sc/trainers/text_text.py:324-378).  keep = 0: the
    recipe's `gradient_checkpointing: true` taken literally (every block recomputed); "auto": selective checkpointing."""
    from contrastors_amd.config import Config, DataArgs, ModelArgs, TrainArgs
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.trainers import TextTextTrainer

    cfg = Config(train_args=TrainArgs(learning_rate=2e-5, weight_decay=0.01, warmup_steps=1, grad_cache=False,
                                      schedule_type="linear", max_grad_norm=1.0, clamp_logits=False,
                                      matryoshka_dims=[768, 512, 256, 128], checkpoint_keep_layers=keep),
                 data_args=DataArgs(batch_size=32, seed=3),
                 model_args=ModelArgs(logit_scale=50.0, pooling="mean", model_name="nomic-embed-text-v1", hamming=True,
                                      num_negatives=7, gradient_checkpointing=True, seq_len=2048))
    tr = TextTextTrainer(cfg, torch.bfloat16, device=dev, trunk_config=NomicBertConfig.nomic_bert_2048(), total_steps=1000)
    g = torch.Generator().manual_seed(8)
    S, nq, nd = 2048, 32, 256
    batch = {}
    for side, n in (("query", nq), ("document", nd)):
        batch[f"{side}_input_ids"] = torch.randint(1000, 30522, (n, S), generator=g).to(dev)
        batch[f"{side}_seqlens"] = [S] * n
    torch.cuda.reset_peak_memory_stats(dev)
    # (selective checkpointing: step 1 takes the recipe literally and measures, step 2 rebuilds the arenas with their kept
    # blocks -- both stay outside the timed window)
    ms = _time_steps(torch, lambda: tr.training_step(batch), 3 if keep == "auto" else 1, max(2, min(steps, 5)))
    tokens = (nq + nd) * S
    fwd = tokens * 12 * _fwd_flop_per_token(768, 2 * 3072 + 3072, S)
    med = _pct(ms, 0.5)
    return {"workload": "configs[2]: nomic-embed-text-v1 finetune, 32 queries + 256 documents x 2048 tokens per GPU, Matryoshka "
                        "{768,512,256,128}, hamming, activation checkpointing, direct step",
            "value": nq / (med * 1e-3), "unit": "query examples/s", "tokens_per_s": tokens / (med * 1e-3), "ms_per_step": med,
            "p10_ms": _pct(ms, 0.1), "p90_ms": _pct(ms, 0.9), "steps": len(ms),
            "roofline": {"bound": "mfma", "achieved": 3 * fwd / (med * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": 3 * fwd / (med * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "executed_frac": 4 * fwd / (med * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                         "note": "algorithmic = 3 forward-equivalents of 302 MFLOP/token (attention = 25 %); executed adds the "
                                 "checkpoint re-forward (all of it when every block is recomputed)"},
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
            "checkpoint_keep_layers": keep,
            "kept_blocks": [{"arena_tokens": t, "kept": k, "of": 12} for t, k in sorted(tr.model["model"].trunk._keep_logged)]}


def leg_image_text(torch, dev, steps, clip: bool, keep=0, hf_dropout=True):
    """configs[3] (LiT: frozen ViT-B/16 image tower, trainable BERT-base text tower) and configs[4] (CLIP: both trained, fp8
    similarity, global batch 32768) at the per-GPU shape of an 8-GPU job: 4096 (image, text) pairs, text seq 77.  On one GPU
    the other seven ranks' gathered embeddings are stand-ins (28672 random unit vectors), so the loss has its real
    4096 x 32768 shape in both directions."""
    import torch.nn.functional as F

    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
    from contrastors_amd.loss import _infonce, _scale_of
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.optimizer import FusedAdamW
    from contrastors_amd.vit import ViTConfig

    n, G, S_t = 4096, 32768, 77
    vis = BiEncoder(BiEncoderConfig(model_name="vit_base_patch16_224", pooling="cls", freeze=not clip,
                                    gradient_checkpointing=clip, checkpoint_keep_layers=keep,
                                    trunk_config=ViTConfig.vit_base_patch16_224()),
                    device=dev, seed=1).train()
    # the text tower as the reference converts bert-base-uncased's hub config: dropout 0.1 on residuals, embeddings and
    # attention probabilities (sc/models/encoder/bert.py:19-21); hf_dropout = False: the architecture alone (sub-record)
    txt = BiEncoder(BiEncoderConfig(model_name="bert-base-uncased", pooling="mean",
                                    trunk_config=NomicBertConfig.bert_base_uncased(hf_dropout=hf_dropout)), device=dev, seed=2).train()
    scale = LogitScale(SimpleNamespace(logit_scale=1 / 0.07, trainable_logit_scale=False)).to(dev)
    towers = [txt] + ([vis] if clip else [])
    groups = [{"params": [], "weight_decay": 0.01}, {"params": [], "weight_decay": 0.0}]
    for t in towers:
        gq = t.param_groups(0.01)
        groups[0]["params"] += gq[0]["params"]
        groups[1]["params"] += gq[1]["params"]
    opt = FusedAdamW(groups, lr=1e-4, betas=(0.9, 0.98), eps=1e-6)
    g = torch.Generator().manual_seed(9)
    pixels = torch.randn(n, 3, 224, 224, generator=g).to(dev).bfloat16()
    t_in = {"input_ids": torch.randint(1000, 30522, (n, S_t), generator=g).to(dev), "seqlens": [S_t] * n}
    others_t = F.normalize(torch.randn(G - n, 768, generator=g), dim=-1).to(dev)
    others_v = F.normalize(torch.randn(G - n, 768, generator=g), dim=-1).to(dev)
    labels = torch.arange(n, device=dev)
    sc, sp = _scale_of(scale)
    loss_ms = []

    def step():
        for t in towers:
            t.trunk.zero_grad()
        v = F.normalize(vis(input_ids=pixels, normalize=False)["embedding"], dim=-1, p=2)
        te = F.normalize(txt(**t_in, normalize=False)["embedding"], dim=-1, p=2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        coef = 0.5 * (G // n) / n
        loss = (_infonce(v, torch.cat([te, others_t]), labels, sc, coef, sp, clip)
                + _infonce(te, torch.cat([v, others_v]), labels, sc, coef, sp, clip))
        e1.record()
        loss.backward()
        opt.step(max_grad_norm=1.0)
        for t in towers:
            t.trunk.sync_shadows()
        loss_ms.append((e0, e1))
        return loss

    torch.cuda.reset_peak_memory_stats(dev)
    ms = _time_steps(torch, step, 3 if keep == "auto" else 1, max(2, min(steps, 5)))
    med = _pct(ms, 0.5)
    f_img = n * 197 * 12 * _fwd_flop_per_token(768, 2 * 3072, 197) + n * 196 * 2.0 * 768 * 768
    f_txt = n * S_t * 12 * _fwd_flop_per_token(768, 2 * 3072, S_t)
    flop = (3 * f_img if clip else f_img) + 3 * f_txt
    lf = sorted(a.elapsed_time(b) for a, b in loss_ms[1:])
    loss_fwd_ms = _pct(lf, 0.5)
    loss_flop = 2 * 2.0 * n * G * 768
    rec = {"workload": ("configs[4]: CLIP-style ViT-B/16 + BERT-base, both towers trained (ViT with activation checkpointing), "
                        "fp8 MFMA similarity, 4096 pairs per GPU against 32768 gathered" if clip else
                        "configs[3]: LiT, frozen ViT-B/16 image tower + trainable BERT-base text tower, 4096 pairs per GPU "
                        "against 32768 gathered, exact fp32 similarity"),
           "value": n / (med * 1e-3), "unit": "pairs/s per GPU", "ms_per_step": med, "p10_ms": _pct(ms, 0.1),
           "p90_ms": _pct(ms, 0.9), "steps": len(ms),
           "roofline": {"bound": "mfma", "achieved": flop / (med * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": flop / (med * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                        "note": "algorithmic encoder FLOPs: image 35.1 GFLOP x " + ("3" if clip else "1 (frozen)") +
                                ", text 13.4 GFLOP x 3 per pair; text tower dropout " + ("0.1 (the reference's "
                                "conversion of the hub config)" if hf_dropout else "0 (architecture only)")},
           "loss_forward": {"ms": loss_fwd_ms, "flop": loss_flop, "achieved": loss_flop / (loss_fwd_ms * 1e-3) / 1e12,
                            "peak": PEAK_FP8_TFLOPS if clip else 157.3, "unit": "TFLOP/s",
                            "frac": loss_flop / (loss_fwd_ms * 1e-3) / 1e12 / (PEAK_FP8_TFLOPS if clip else 157.3),
                            "kernel": "infonce_fp8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4) + row quantisation" if clip else
                                      "sgemm_nt_kernel<LSE> (v_mfma_f32_32x32x2_f32, exact)",
                            "note": "both directions of the 4096 x 32768 x 768 similarity + online log-sum-exp"},
           "text_tower_dropout": 0.1 if hf_dropout else 0.0,
           "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}
    if clip:
        rec["checkpoint_keep_layers"] = keep
        rec["kept_blocks"] = [{"arena_tokens": t, "kept": k, "of": 12} for t, k in sorted(vis.trunk._keep_logged)]
    return rec


def run_config_legs(torch, dev, steps):
    import gc

    out = {}
    for name, fn in (("cfg1", lambda: leg_cfg1(torch, dev, steps)), ("cfg3", lambda: leg_cfg3(torch, dev, steps)),
                     ("lit", lambda: leg_image_text(torch, dev, steps, clip=False)),
                     ("clip", lambda: leg_image_text(torch, dev, steps, clip=True))):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- a secondary record must never take the headline down
            out[name] = f"failed: {type(e).__name__}: {e}"[:300]
        gc.collect()
        torch.cuda.empty_cache()
    # the three legs with a BERT-base tower once more with dropout 0 (rounds 1-3 reported only this; the records above are the
    # reference's recipe): what the Philox masks cost
    for name, fn in (("cfg1", lambda: leg_cfg1(torch, dev, steps, hf_dropout=False)),
                     ("lit", lambda: leg_image_text(torch, dev, steps, clip=False, hf_dropout=False)),
                     ("clip", lambda: leg_image_text(torch, dev, steps, clip=True, hf_dropout=False))):
        if not isinstance(out.get(name), dict):
            continue
        try:
            r = fn()
            out[name]["dropout_0"] = {k: r[k] for k in ("value", "unit", "ms_per_step", "p10_ms", "p90_ms", "steps") if k in r}
            out[name]["dropout_0"]["frac_of_mfma_peak"] = r["roofline"]["frac"] if "roofline" in r else r.get("frac_of_mfma_peak")
        except Exception as e:  # noqa: BLE001
            out[name]["dropout_0"] = f"failed: {type(e).__name__}: {e}"[:300]
        gc.collect()
        torch.cuda.empty_cache()
    # the two legs whose recipes switch activation checkpointing on, once more with selective checkpointing ("auto")
    for name, fn in (("cfg3", lambda: leg_cfg3(torch, dev, steps, keep="auto")),
                     ("clip", lambda: leg_image_text(torch, dev, steps, clip=True, keep="auto"))):
        if not isinstance(out.get(name), dict):
            continue
        try:
            out[name]["selective_checkpointing"] = _selective(fn())
        except Exception as e:  # noqa: BLE001
            out[name]["selective_checkpointing"] = f"failed: {type(e).__name__}: {e}"[:300]
        gc.collect()
        torch.cuda.empty_cache()
    return out


def flat_scalars(extra: dict) -> dict:
    """The secondary records once more as flat scalar keys: the driver's `parsed` view of the JSON line keeps scalars, the
    nested records only by name."""
    f = {}

    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d if isinstance(d, (int, float)) else None

    for leg, unit_key in (("cfg1", "cfg1_pairs_s"), ("cfg3", "cfg3_query_examples_s"), ("lit", "lit_pairs_s_per_gpu"),
                          ("clip", "clip_pairs_s_per_gpu")):
        f[unit_key] = get(extra, leg, "value")
        f[f"{leg}_frac"] = get(extra, leg, "roofline", "frac") if leg != "cfg1" else get(extra, leg, "frac_of_mfma_peak")
        f[f"{leg}_ms_per_step"] = get(extra, leg, "ms_per_step")
    for leg in ("cfg1", "lit", "clip"):
        f[f"{leg}_dropout0_value"] = get(extra, leg, "dropout_0", "value")
    for leg in ("cfg3", "clip"):
        f[f"{leg}_selective_value"] = get(extra, leg, "selective_checkpointing", "value")
        f[f"{leg}_selective_frac"] = get(extra, leg, "selective_checkpointing", "frac_of_mfma_peak")
        f[f"{leg}_selective_p10_ms"] = get(extra, leg, "selective_checkpointing", "p10_ms")
        f[f"{leg}_selective_p90_ms"] = get(extra, leg, "selective_checkpointing", "p90_ms")
    f["fp8_loss_frac_of_fp8_peak"] = get(extra, "clip", "loss_forward", "frac")
    f["weak_pairs_s"] = get(extra, "weak", "value")
    f["resident_pairs_s"] = get(extra, "resident", "value")
    f["auto_policy_pairs_s"] = get(extra, "auto_policy", "value")
    f["dropin_chunk64_pairs_s"] = get(extra, "dropin_chunk64", "value")
    f["exact_chunk64_pairs_s"] = get(extra, "dropin_chunk64", "exact_chunk64", "value")
    return {k: v for k, v in f.items() if v is not None}



def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: spawn the N ranks ourselves (same command line the
    driver would use) and pass their output through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CX_BENCH_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.seq_len)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test hook (tests/test_distributed_gpu.py): several ranks may share one GPU with gloo carrying the device tensors;
    # the product launch is always one rank per GPU over RCCL ("nccl").
    backend = os.environ.get("CX_BENCH_BACKEND", "nccl")
    if os.environ.get("CX_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from contrastors_amd import _C
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
    from contrastors_amd.distributed import exchange_report, set_exchange_timeout
    from contrastors_amd.loss import grad_cache_loss
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.optimizer import FusedAdamW
    from contrastors_amd.policy import GradCachePolicy

    if args.only_config_legs:   # (rocprofv3 of one secondary leg: scripts/gpu_r3_final.sh)
        want = set(args.only_config_legs.split(","))
        fns = {"cfg1": lambda: leg_cfg1(torch, dev, args.steps), "cfg1_nodrop": lambda: leg_cfg1(torch, dev, args.steps, hf_dropout=False),
               "lit_nodrop": lambda: leg_image_text(torch, dev, args.steps, False, hf_dropout=False),
               "cfg3": lambda: leg_cfg3(torch, dev, args.steps),
               "lit": lambda: leg_image_text(torch, dev, args.steps, False), "clip": lambda: leg_image_text(torch, dev, args.steps, True),
               "cfg3_selective": lambda: leg_cfg3(torch, dev, args.steps, keep="auto"),
               "clip_selective": lambda: leg_image_text(torch, dev, args.steps, True, keep="auto")}
        import gc

        recs = {}
        for k in fns:
            if k in want:
                recs[k] = fns[k]()
                gc.collect()
                torch.cuda.empty_cache()
        print(json.dumps(recs), flush=True)
        return
    set_exchange_timeout(20.0)   # ranks of a benchmark arrive together: a peer 20 s late is a failed exchange (-> process group), not a stall
    for k in ("CX_GRADCACHE_CHUNK", "CX_GRADCACHE_RESIDENT"):   # the legs below state their schedule as config, not environment
        os.environ.pop(k, None)
    # --chunk-size is taken literally in every leg but the drop-in one; the metric is the two-pass GradCache step (the
    # `resident` record is separate)
    POL = {"metric": GradCachePolicy(chunk="exact", resident=False), "resident": GradCachePolicy(chunk="exact", resident=True),
           "dropin": GradCachePolicy(chunk="auto", resident=False), "auto": GradCachePolicy(chunk="exact", resident="auto")}
    lib = _C.lib()
    G, S = args.global_batch, args.seq_len
    assert G % world == 0
    cfg = NomicBertConfig.nomic_bert_2048(n_layer=args.layers)
    tower = BiEncoder(BiEncoderConfig(model_name="nomic-ai/nomic-bert-2048", pooling="mean", logit_scale=50.0,
                                      trunk_config=cfg), device=dev, seed=0).train()
    tower.broadcast_parameters(0)
    scale = LogitScale(SimpleNamespace(logit_scale=50.0, trainable_logit_scale=False)).to(dev)
    opt = FusedAdamW(tower.param_groups(0.01), lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    total_steps, warm = 12000, 700  # contrastive_pretrain.yaml: cosine, warmup 700
    sched = torch.optim.lr_scheduler.LambdaLR(
        opt, lambda s: (s + 1) / warm if s < warm else 0.5 * (1 + math.cos(math.pi * (s - warm) / (total_steps - warm))))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(pairs_per_gpu: int, chunk: int, steps: int, warmup: int, prof: bool, policy=None, step_ms=None):
        """W untimed + K timed steps at `pairs_per_gpu` pairs on every rank; returns the max-over-ranks wall time.  step_ms (a
        list) receives the per-step durations of this rank from events recorded between the steps (no synchronisation
        inside the timed region)."""
        policy = policy or POL["metric"]
        # synthetic (query, document) token ids, SURVEY.md §8(d): staged on the device before timing
        g = torch.Generator().manual_seed(1234 + rank)
        q_ids = torch.randint(1000, 30522, (pairs_per_gpu, S), generator=g)
        d_ids = torch.randint(1000, 30522, (pairs_per_gpu, S), generator=g)
        q_ids[:, 0] = 101
        d_ids[:, 0] = 101
        lens = [S] * pairs_per_gpu
        q_in = {"input_ids": q_ids.to(dev), "seqlens": lens}
        d_in = {"input_ids": d_ids.to(dev), "seqlens": lens}

        def step():
            tower.trunk.zero_grad()
            loss = grad_cache_loss(tower, q_in, tower, d_in, chunk, scale, policy=policy)
            opt.step(max_grad_norm=1.0)  # global-norm clip + AdamW fused (cx_grad_sq_norm + cx_adamw_clip_step)
            sched.step()
            tower.trunk.sync_shadows()
            return loss

        for _ in range(warmup):
            step()
        fence()
        if prof:
            lib.cx_prof_gemm_config(1, args.prof_stride)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if step_ms is not None else None
        t0 = time.perf_counter()
        if marks:
            marks[0].record()
        for i in range(steps):
            loss = step()
            if marks:
                marks[i + 1].record()
        fence()
        dt = time.perf_counter() - t0
        if marks:
            step_ms.extend(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), float(loss.item())

    # ---- what THIS box sustains, measured right before the metric: register-only bf16 MFMA loop + HBM copy
    box = {}
    if not args.no_calibration:
        try:
            from scripts.box_calibration import SmiSampler, calibrate

            box = calibrate(dev)
        except Exception as e:  # noqa: BLE001 -- calibration must never take the headline down
            box = {"error": f"{type(e).__name__}: {e}"[:200]}
            SmiSampler = None
    else:
        SmiSampler = None
    # ---- the metric: global batch 16384 (strong scaling) -------------------------------------------------------------
    b = G // world
    step_ms = []
    tower.exposed_reduce_marks = [] if world > 1 else None
    sampler = SmiSampler(local_rank).start() if SmiSampler is not None else None   # socket power + shader clock DURING the timed region
    dt, loss_last = run_leg(b, args.chunk_size, args.steps, args.warmup, prof=True, step_ms=step_ms)
    if sampler is not None:
        box.update(sampler.stop())
    from contrastors_amd import loss as cx_loss

    sched_local = dict(cx_loss.LAST_SCHEDULE)
    exposed_ms = None
    if world > 1:
        marks = tower.exposed_reduce_marks[-args.steps:]
        tower.exposed_reduce_marks = None
        mine = sorted(a.elapsed_time(c) for a, c in marks) if marks else []
        t = torch.tensor([_pct(mine, 0.5) if mine else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exposed_ms = float(t.item())
        scheds = [None] * world
        dist.all_gather_object(scheds, sched_local)
    else:
        scheds = [sched_local]
    ms, fl = C.c_double(), C.c_double()
    n_t, n_all = C.c_long(), C.c_long()
    lib.cx_prof_gemm_collect(C.byref(ms), C.byref(fl), C.byref(n_t), C.byref(n_all))
    lib.cx_prof_gemm_config(0, 1)
    peak_hbm = torch.cuda.max_memory_allocated(dev)

    extra = {}
    if not args.no_extra_legs:
        few = max(2, args.steps // 4)
        # weak-scaling curve: 2048 pairs per GPU whatever N is (coincides with the strong point at N = 8)
        if b != WEAK_PAIRS_PER_GPU and G >= WEAK_PAIRS_PER_GPU:
            wdt, _ = run_leg(WEAK_PAIRS_PER_GPU, args.chunk_size, args.steps, 1, prof=False)
            extra["weak"] = {"value": WEAK_PAIRS_PER_GPU * world * args.steps / wdt, "unit": "pairs/s",
                             "per_gpu": WEAK_PAIRS_PER_GPU * args.steps / wdt,   # / the N = 1 run's `weak.value` = weak-scaling efficiency
                             "pairs_per_gpu": WEAK_PAIRS_PER_GPU, "global_batch": WEAK_PAIRS_PER_GPU * world,
                             "ms_per_step": 1e3 * wdt / args.steps, "steps": args.steps, "scaling": "weak"}
        elif b == WEAK_PAIRS_PER_GPU:
            extra["weak"] = "same point as the headline (2048 pairs per GPU)"
        # the same 2048 pairs per GPU with the activations of pass 1 kept in HBM (193 GB of 288): nothing to recompute in
        # pass 2, identical loss and gradients (tests/test_loss_gpu.py), 3 forward-equivalents of FLOPs instead of 4
        if G >= WEAK_PAIRS_PER_GPU:
            try:
                # (the headline's chunk-4096 arena -- 167 GB -- is twice what a 2048-pair side needs: pooled, it would sit beside the two
                # resident arenas of this leg at ~270 of 288 GB; a training run with this schedule only ever holds ITS arenas)
                tower.trunk.drop_idle_arenas()
                torch.cuda.empty_cache()
                rdt, _ = run_leg(WEAK_PAIRS_PER_GPU, args.chunk_size, args.steps, 1, prof=False, policy=POL["resident"])
                extra["resident"] = {"value": WEAK_PAIRS_PER_GPU * world * args.steps / rdt, "unit": "pairs/s",
                                     "pairs_per_gpu": WEAK_PAIRS_PER_GPU, "global_batch": WEAK_PAIRS_PER_GPU * world,
                                     "ms_per_step": 1e3 * rdt / args.steps, "steps": args.steps,
                                     "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                                     "note": "GradCache with pass 1's activations resident (train_args.gradcache_resident: auto, the "
                                             "library default, picks this whenever the per-GPU batch fits -- and falls back to two "
                                             "passes on an out-of-memory error: 8 GPUs x 2048 "
                                             "pairs is the metric's own per-GPU shape); same loss bit for bit, gradients equal up to fp32-atomics order, "
                                             "encoder FLOPs 3/4 of the two-pass step. NOT the headline: `value` stays two-pass"}
            except torch.OutOfMemoryError:
                extra["resident"] = "activations of 2048 pairs per GPU did not fit beside this run's other buffers"
        # what an unmodified reference YAML gets: GradCache chunk_size 64 (8192 token rows per GEMM launch)
        if world == 1 and args.chunk_size != 64:
            nb = min(b, 2048)
            cdt, _ = run_leg(nb, 64, few, 1, prof=False)                         # chunk_size 64 taken literally: 8192 token rows per launch
            adt, _ = run_leg(nb, 64, few, 1, prof=False, policy=POL["dropin"])   # the default: the recipe's 64 is a lower bound on a 288 GB part
            extra["dropin_chunk64"] = {"value": nb * few / adt, "unit": "pairs/s", "recipe_chunk_size": 64,
                                       "global_batch": nb, "ms_per_step": 1e3 * adt / few, "steps": few,
                                       "exact_chunk64": {"value": nb * few / cdt, "ms_per_step": 1e3 * cdt / few},
                                       "note": "reference recipe chunk_size 64 (contrastive_pretrain.yaml:15): `value` is what "
                                               "an unmodified YAML gets (train_args.gradcache_chunk: auto raises the chunk to ~262144 "
                                               "tokens, results unchanged), exact_chunk64 = the literal 64 (gradcache_chunk: "
                                               "exact); loss rows x 2048 documents, encoder work per pair unchanged"}
        # the metric's own step under the library's DEFAULT policy (train_args.gradcache_resident: auto): the whole batch does
        # not fit at N = 1 (1.26 TB of saved activations), so the tail of the batch that does fit keeps its activations and
        # skips the re-forward (loss.resident_tail_plan; same loss bit for bit, gradients to fp32 summation order).  Runs
        # LAST among this tower's legs: it leaves ~250 GB of arenas in the engine's pool.
        if world == 1 and b >= 4 * args.chunk_size:
            try:
                tower.trunk.drop_idle_arenas()   # (a training run with this schedule only ever holds ITS arenas)
                torch.cuda.reset_peak_memory_stats(dev)
                a_ms = []
                adt2, _ = run_leg(b, args.chunk_size, max(2, args.steps // 2), 1, prof=False, policy=POL["auto"], step_ms=a_ms)
                n_chunks = (b + args.chunk_size - 1) // args.chunk_size
                extra["auto_policy"] = {"value": b * len(a_ms) / adt2, "unit": "pairs/s", "global_batch": b,
                                        "ms_per_step": 1e3 * adt2 / len(a_ms), "steps": len(a_ms),
                                        "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                                        "chunks_per_side": n_chunks,
                                        "note": "same step, train_args.gradcache_resident: auto (the library default) instead of the "
                                                "headline's literal two-pass schedule: the document-side tail chunks that fit in "
                                                "HBM beside the no-grad arena keep their activations in pass 1 and are "
                                                "back-propagated first in pass 2 without a re-forward (their arenas then serve "
                                                "the re-forwards); identical loss, gradients to fp32 summation order "
                                                "(tests/test_loss_gpu.py::test_partially_resident_gradcache_equals_two_pass). "
                                                "NOT the headline: `value` stays the reference's two passes for every chunk"}
            except torch.OutOfMemoryError:
                extra["auto_policy"] = "the kept tail did not fit beside this run's other buffers"
        # the loss path's one exchange step on its own: all-gather of (16384 / N, 768) fp32 embeddings per rank, through the
        # process group (RCCL) and through the one-shot peer-store path (csrc/xgmi.hip), against 7 x 153 GB/s of xGMI per GPU
        if world > 1:
            emb = torch.randn(b, cfg.n_embd, device=dev)
            recv = (world - 1) * emb.numel() * 4

            def time_gather(fn, reps=50):
                for _ in range(5):
                    fn(emb)
                fence()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn(emb)
                e1.record()
                torch.cuda.synchronize()
                t = torch.tensor([e0.elapsed_time(e1) / reps * 1e-3], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())

            def record(seconds, what):
                return {"bytes_received_per_gpu": recv, "seconds": seconds, "achieved": recv / seconds / 1e9,
                        "peak": XGMI_PEAK_GBS, "unit": "GB/s", "frac": recv / seconds / 1e9 / XGMI_PEAK_GBS, "collective": what}

            rec = {}
            # what the data path ITSELF decided at its first exchange (warm-up step): set-up, bit-exact check against the
            # process group, race at the step's payload, faster one taken on every rank (contrastors_amd/distributed.py)
            extra["exchange"] = exchange_report()
            rec["process_group"] = record(time_gather(lambda t: dist.all_gather_into_tensor(
                torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device), t)),
                f"{backend} all_gather_into_tensor, one fused buffer in rank order" + (" (RCCL over xGMI)" if backend == "nccl" else ""))
            try:
                from contrastors_amd import distributed as cxd

                ex = cxd._ONESHOT   # the exchange the selection set up and verified (None if it was unavailable)
                if ex is None:
                    raise RuntimeError(str(extra["exchange"].get("reason", "not set up")))
                rec["oneshot"] = record(time_gather(ex.all_gather), "one-shot: every rank stores its shard into every peer's IPC "
                                                                   "buffer + one system-scope flag exchange + copy-out of the "
                                                                   "receive buffer")
                rec["oneshot_in_place"] = record(time_gather(lambda t: ex.all_gather(t, copy=False)),
                                                 "the same exchange, result read in the receive buffer (no copy-out)")
                ex.check()
            except Exception as e:  # noqa: BLE001 -- the record must never take the benchmark down
                rec["oneshot"] = f"unavailable: {type(e).__name__}: {e}"[:300]
            rec["carried_by"] = extra["exchange"].get("choice")
            rec["note"] = ("the step's gather_with_grad runs on `carried_by` (chosen by measurement at start-up, "
                           "train_args.exchange: auto)" + ("; under the shared-GPU test backend the numbers are not xGMI numbers"
                                                          if backend != "nccl" else ""))
            extra["xgmi_allgather"] = rec
    if world == 1 and not args.no_extra_legs and not args.no_config_legs:
        # the other BASELINE configs at their per-GPU shapes (the metric's tower is released first: they need the HBM)
        import gc

        del tower, opt, sched
        gc.collect()
        torch.cuda.empty_cache()
        extra.update(run_config_legs(torch, dev, args.steps))
    if rank == 0:
        # HBM bytes per GEMM launch: not measurable from inside the process; taken from the committed rocprofv3 PMC
        # passes of this same command line (scripts/gpu_round.sh PMC=1 -> scripts/pmc_traffic.py) when they were
        # collected at the same launch sizes (same GradCache chunk), else null.
        traffic, traffic_src = None, None
        for name in ("r6_pmc_gemm_traffic.json", "r5_pmc_gemm_traffic.json", "r4_pmc_gemm_traffic.json", "r3_pmc_gemm_traffic.json", "r2_pmc_gemm_traffic.json", "r1_pmc_gemm_traffic.json"):
            try:
                tj = json.load(open(ROOT / "profiles" / name))
                if tj.get("grad_cache_chunk") == min(args.chunk_size, b):
                    traffic, traffic_src = tj["hbm_bytes_per_launch"], f"profiles/{name}"
                    break
            except (OSError, ValueError, KeyError):
                pass
        pairs_per_s = G * args.steps / dt
        achieved = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        probe = box.get("mfma_probe_tflops") if isinstance(box, dict) else None
        out = {
            "metric": "query-doc pairs/sec (whole node), nomic-bert-2048 seq128 global-batch 16384",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            # per-step durations of rank 0 (events between the steps of the same timed region; SURVEY.md §8(d): median + p10 / p90)
            "step_ms": {"median": _pct(sorted(step_ms), 0.5), "p10": _pct(sorted(step_ms), 0.1), "p90": _pct(sorted(step_ms), 0.9),
                        "min": min(step_ms), "max": max(step_ms), "n": len(step_ms)},
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: nomic-bert-2048 bi-encoder contrastive pretrain step (GradCache, "
                                   "paired InfoNCE scale 50, AdamW, clip 1.0)",
                       "global_batch": G, "pairs_per_gpu": b, "seq_len": S, "grad_cache_chunk": args.chunk_size,
                       "n_layer": cfg.n_layer, "parallelism": f"dp{world}", "loss_last_step": loss_last,
                       "peak_hbm_gb": round(peak_hbm / 2**30, 1),
                       "launch": "self (torch.distributed.run)" if os.environ.get("CX_BENCH_SELF_LAUNCHED") else
                                 ("torch.distributed.run" if world > 1 else "single process")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS,
                         # the same achieved rate against what this box's matrix cores sustain at its power limit (box.mfma_probe_tflops,
                         # measured seconds before the timed region): the number to compare ACROSS boxes
                         "frac_of_box_ceiling": (achieved / probe) if probe else None,
                         "traffic": traffic, "traffic_unit": "B/launch",
                         "traffic_source": traffic_src,
                         "kernel": "bf16 GEMM family: gemm_bf16_v6_kernel (fwd, dgrad, fused SwiGLU fc1) + "
                                   "gemm_bf16_v6tn_kernel (wgrad)",
                         "launches_timed": n_t.value, "launches_total": n_all.value,
                         "avg_launch_us": 1e3 * ms.value / max(1, n_t.value),
                         "algorithmic_flop_per_launch": fl.value / max(1, n_t.value),
                         "whole_step_frac_of_mfma_peak": pairs_per_s / world * GFLOP_PER_PAIR / 1e3 / PEAK_BF16_TFLOPS},
        }
        out.update(extra)
        out.update(flat_scalars(extra))
        out["whole_step_frac_of_mfma_peak"] = out["roofline"]["whole_step_frac_of_mfma_peak"]
        # per-box calibration (flat scalars: the driver's `parsed` view keeps scalars) + the nested record
        out["box"] = box
        for k in ("mfma_probe_tflops", "blas_ref_tflops", "v6_same_shapes_tflops", "hbm_copy_tbs", "mean_sclk_mhz", "mean_power_w", "power_cap_w", "mfma_probe_clock_mhz", "mfma_probe_power_w"):
            if isinstance(box, dict) and isinstance(box.get(k), (int, float)):
                out[f"box_{k}"] = box[k]
        for k in ("like_for_like", "like_for_like_short_k", "like_for_like_long_k"):   # plain v6 / vendor BLAS on the calibration's two shapes, interleaved windows (round 6)
            if isinstance(box, dict) and isinstance(box.get(k), (int, float)):
                out[k] = box[k]
        blas_ref = box.get("blas_ref_tflops") if isinstance(box, dict) else None
        if blas_ref:   # the shipped GEMM family against the vendor BLAS on the same box (two of the step's shapes): steadier across boxes than the probe
            out["vs_box_blas"] = achieved / blas_ref
            out["roofline"]["vs_box_blas"] = achieved / blas_ref
        if probe:
            out["frac_of_box_ceiling"] = achieved / probe
            out["whole_step_frac_of_box_ceiling"] = pairs_per_s / world * GFLOP_PER_PAIR / 1e3 / probe
        # the schedule every rank's metric leg actually ran (resident / partial / two-pass) and, for N > 1, the part of the
        # gradient reduction nothing overlapped (end of the last backward kernel -> reduced gradient, median over steps, max over ranks)
        out["schedule_per_rank"] = scheds
        out["schedule"] = scheds[0].get("schedule") if scheds and isinstance(scheds[0], dict) else None
        # The metric's FLOP count is the reference's schedule: 4 forward-equivalents per pair (no-grad forward, re-forward, dgrad, wgrad).  A
        # resident / partially resident schedule (the library's default policy when the per-rank batch fits: at 2048 pairs per GPU it does) skips
        # the re-forward of the kept sequences: `executed_flop_frac` = executed / algorithmic FLOPs of rank 0's schedule, and
        # `whole_step_frac_executed` = the roofline fraction on what was actually computed (without it the N = 8 fraction
        # of a resident run reads 4/3 too high).  1.0 on the headline leg (two-pass by construction).
        s0 = scheds[0] if scheds and isinstance(scheds[0], dict) else {}
        tot_seqs = 2.0 * (G // world)
        kept = float(s0.get("kept_q_seqs") or 0) + float(s0.get("kept_d_seqs") or 0)
        if s0.get("schedule") == "resident":
            kept = tot_seqs
        out["executed_flop_frac"] = (3.0 + (1.0 - min(1.0, kept / tot_seqs))) / 4.0 if tot_seqs > 0 else None
        if out["executed_flop_frac"]:
            out["whole_step_frac_executed"] = out["whole_step_frac_of_mfma_peak"] * out["executed_flop_frac"]
        if exposed_ms is not None:
            out["allreduce_exposed_ms"] = exposed_ms
        xg = extra.get("xgmi_allgather") if isinstance(extra.get("xgmi_allgather"), dict) else {}
        for tag, key in (("rccl", "process_group"), ("oneshot", "oneshot"), ("oneshot_in_place", "oneshot_in_place")):
            r = xg.get(key)
            if isinstance(r, dict):
                out[f"allgather_us_{tag}"] = 1e6 * r["seconds"]
                out[f"allgather_xgmi_frac_{tag}"] = r["frac"]
        if isinstance(extra.get("weak"), dict):
            out["weak_pairs_s_per_gpu"] = extra["weak"]["per_gpu"]
        elif world > 1 and b == WEAK_PAIRS_PER_GPU:
            out["weak_pairs_s_per_gpu"] = pairs_per_s / world
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(S)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
