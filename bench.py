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
  xgmi_allgather  (N > 1) the embedding all-gather of the loss timed on its own, against 7 x 153 GB/s of xGMI per GPU
  cpu_baseline  (N = 1) oracle = CPU restatement of the reference, one 64-pair chunk forward + backward, min of 3
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
    ap.add_argument("--chunk-size", type=int, default=int(os.environ.get("CX_BENCH_CHUNK", 2048)),
                    help="GradCache chunk = sequences per encoder call.  A pure memory knob (results are identical); the "
                         "reference recipe uses 64 on 80 GB parts (contrastive_pretrain.yaml:15); 2048 = 262144 token rows per "
                         "GEMM launch (1024 whole 256-row tile panels) and peaks at ~115 GB of the MI355X's 288 GB")
    ap.add_argument("--layers", type=int, default=12, help=argparse.SUPPRESS)  # debugging only; 12 = the metric
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the weak-scaling and chunk-64 records")
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
    return {"value": pairs / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"one {pairs}-pair chunk x seq {seq_len}, direct fwd+bwd+InfoNCE (3x fwd FLOPs, no GradCache "
                      f"re-forward), fp32 torch-CPU oracle port (the reference tree is absent on the GPU box), min of "
                      f"{reps} reps = {dt:.2f} s (all: {', '.join(f'{t:.2f}' for t in times)})"}


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
    from contrastors_amd.distributed import gather_with_grad
    from contrastors_amd.loss import grad_cache_loss
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.optimizer import FusedAdamW

    os.environ["CX_GRADCACHE_CHUNK"] = "exact"  # --chunk-size is taken literally in every leg but the drop-in one
    os.environ["CX_GRADCACHE_RESIDENT"] = "0"   # the metric is the two-pass GradCache step; the `resident` record is separate
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

    def run_leg(pairs_per_gpu: int, chunk: int, steps: int, warmup: int, prof: bool):
        """W untimed + K timed steps at `pairs_per_gpu` pairs on every rank; returns the max-over-ranks wall time."""
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
            loss = grad_cache_loss(tower, q_in, tower, d_in, chunk, scale)
            opt.step(max_grad_norm=1.0)  # global-norm clip + AdamW fused (cx_grad_sq_norm + cx_adamw_clip_step)
            sched.step()
            tower.trunk.sync_shadows()
            return loss

        for _ in range(warmup):
            step()
        fence()
        if prof:
            lib.cx_prof_gemm_config(1, args.prof_stride)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), float(loss.item())

    # ---- the metric: global batch 16384 (strong scaling) -------------------------------------------------------------
    b = G // world
    dt, loss_last = run_leg(b, args.chunk_size, args.steps, args.warmup, prof=True)
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
                             "pairs_per_gpu": WEAK_PAIRS_PER_GPU, "global_batch": WEAK_PAIRS_PER_GPU * world,
                             "ms_per_step": 1e3 * wdt / args.steps, "steps": args.steps, "scaling": "weak"}
        elif b == WEAK_PAIRS_PER_GPU:
            extra["weak"] = "same point as the headline (2048 pairs per GPU)"
        # the same 2048 pairs per GPU with the activations of pass 1 kept in HBM (193 GB of 288): nothing to recompute in
        # pass 2, identical loss and gradients (tests/test_loss_gpu.py), 3 forward-equivalents of FLOPs instead of 4
        if G >= WEAK_PAIRS_PER_GPU:
            os.environ["CX_GRADCACHE_RESIDENT"] = "1"
            try:
                rdt, _ = run_leg(WEAK_PAIRS_PER_GPU, args.chunk_size, args.steps, 1, prof=False)
                extra["resident"] = {"value": WEAK_PAIRS_PER_GPU * world * args.steps / rdt, "unit": "pairs/s",
                                     "pairs_per_gpu": WEAK_PAIRS_PER_GPU, "global_batch": WEAK_PAIRS_PER_GPU * world,
                                     "ms_per_step": 1e3 * rdt / args.steps, "steps": args.steps,
                                     "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
                                     "note": "GradCache with pass 1's activations resident (CX_GRADCACHE_RESIDENT=auto, the "
                                             "library default, picks this whenever the per-GPU batch fits: 8 GPUs x 2048 "
                                             "pairs is the metric's own per-GPU shape); same loss bit for bit, gradients equal up to fp32-atomics order, "
                                             "encoder FLOPs 3/4 of the two-pass step. NOT the headline: `value` stays two-pass"}
            except torch.OutOfMemoryError:
                extra["resident"] = "activations of 2048 pairs per GPU did not fit beside this run's other buffers"
            os.environ["CX_GRADCACHE_RESIDENT"] = "0"
        # what an unmodified reference YAML gets: GradCache chunk_size 64 (8192 token rows per GEMM launch)
        if world == 1 and args.chunk_size != 64:
            nb = min(b, 2048)
            os.environ["CX_GRADCACHE_CHUNK"] = "exact"   # chunk_size 64 taken literally: 8192 token rows per launch
            cdt, _ = run_leg(nb, 64, few, 1, prof=False)
            os.environ["CX_GRADCACHE_CHUNK"] = "auto"    # the default: the recipe's 64 is a lower bound on a 288 GB part
            adt, _ = run_leg(nb, 64, few, 1, prof=False)
            os.environ["CX_GRADCACHE_CHUNK"] = "exact"
            extra["dropin_chunk64"] = {"value": nb * few / adt, "unit": "pairs/s", "recipe_chunk_size": 64,
                                       "global_batch": nb, "ms_per_step": 1e3 * adt / few, "steps": few,
                                       "exact_chunk64": {"value": nb * few / cdt, "ms_per_step": 1e3 * cdt / few},
                                       "note": "reference recipe chunk_size 64 (contrastive_pretrain.yaml:15): `value` is what "
                                               "an unmodified YAML gets (CX_GRADCACHE_CHUNK=auto raises the chunk to ~131072 "
                                               "tokens, results unchanged), exact_chunk64 = the literal 64 (CX_GRADCACHE_CHUNK="
                                               "exact); loss rows x 2048 documents, encoder work per pair unchanged"}
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
            if backend == "nccl":
                rec["rccl"] = record(time_gather(gather_with_grad), "RCCL all_gather_into_tensor, one fused buffer in rank order")
            try:
                from contrastors_amd.distributed import OneShotExchange

                ex = OneShotExchange(world * emb.numel() * 4, device=dev)
                got = ex.all_gather(emb)
                ref = torch.empty_like(got)
                dist.all_gather_into_tensor(ref, emb)
                same = torch.tensor([float(torch.equal(got, ref))], device=dev)
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                if float(same.item()) != 1.0:
                    raise RuntimeError("one-shot all-gather disagrees with the process group's all-gather")
                rec["oneshot"] = record(time_gather(ex.all_gather), "one-shot: every rank stores its shard into every peer's IPC "
                                                                   "buffer + one system-scope flag exchange + copy-out of the "
                                                                   "receive buffer (what CX_EXCHANGE=oneshot runs)")
                rec["oneshot_in_place"] = record(time_gather(lambda t: ex.all_gather(t, copy=False)),
                                                 "the same exchange, result read in the receive buffer (no copy-out)")
                ex.check()
                ex.close()
            except Exception as e:  # noqa: BLE001 -- the record must never take the benchmark down
                rec["oneshot"] = f"unavailable: {type(e).__name__}: {e}"[:300]
            if rec:
                rec["note"] = ("the data path uses the process group's collectives unless CX_EXCHANGE=oneshot; under the shared-GPU "
                               "test backend the numbers are not xGMI numbers") if backend != "nccl" else \
                    "the data path uses the process group's collectives unless CX_EXCHANGE=oneshot"
                extra["xgmi_allgather"] = rec
    if rank == 0:
        # HBM bytes per GEMM launch: not measurable from inside the process; taken from the committed rocprofv3 PMC
        # passes of this same command line (scripts/gpu_round.sh PMC=1 -> scripts/pmc_traffic.py) when they were
        # collected at the same launch sizes (same GradCache chunk), else null.
        traffic, traffic_src = None, None
        for name in ("r2_pmc_gemm_traffic.json", "r1_pmc_gemm_traffic.json"):
            try:
                tj = json.load(open(ROOT / "profiles" / name))
                if tj.get("grad_cache_chunk") == min(args.chunk_size, b):
                    traffic, traffic_src = tj["hbm_bytes_per_launch"], f"profiles/{name}"
                    break
            except (OSError, ValueError, KeyError):
                pass
        pairs_per_s = G * args.steps / dt
        achieved = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        out = {
            "metric": "query-doc pairs/sec (whole node), nomic-bert-2048 seq128 global-batch 16384",
            "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[1]: nomic-bert-2048 bi-encoder contrastive pretrain step (GradCache, "
                                   "paired InfoNCE scale 50, AdamW, clip 1.0)",
                       "global_batch": G, "pairs_per_gpu": b, "seq_len": S, "grad_cache_chunk": args.chunk_size,
                       "n_layer": cfg.n_layer, "parallelism": f"dp{world}", "loss_last_step": loss_last,
                       "peak_hbm_gb": round(peak_hbm / 2**30, 1),
                       "launch": "self (torch.distributed.run)" if os.environ.get("CX_BENCH_SELF_LAUNCHED") else
                                 ("torch.distributed.run" if world > 1 else "single process")},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_unit": "B/launch",
                         "traffic_source": traffic_src,
                         "kernel": "bf16 GEMM family: gemm_bf16_v6_kernel (fwd, dgrad, fused SwiGLU fc1) + "
                                   "gemm_bf16_v6tn_kernel (wgrad)",
                         "launches_timed": n_t.value, "launches_total": n_all.value,
                         "avg_launch_us": 1e3 * ms.value / max(1, n_t.value),
                         "algorithmic_flop_per_launch": fl.value / max(1, n_t.value),
                         "whole_step_frac_of_mfma_peak": pairs_per_s / world * GFLOP_PER_PAIR / 1e3 / PEAK_BF16_TFLOPS},
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(S)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
