"""N>1 data path with the real HIP kernels: two ranks share the one GPU of the test box (gloo carries the device
tensors; the production launch is one rank per GPU over RCCL, which needs two GPUs).  What must hold (sc/loss.py:76-132,
sc/distributed.py:5-12, DDP's gradient averaging): the rank-sharded GradCache step leaves, on every rank, W x the
gradient of the single-process step over the same global batch (each rank's loss is CE x W and the reduction averages),
and the rank losses average to W x the single-process loss."""
import json
import os
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _tower(dev):
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
    from contrastors_amd.nomic_bert import NomicBertConfig
    from oracle.make_golden import TINY_NOMIC

    tc = NomicBertConfig(**{k: v for k, v in TINY_NOMIC.items() if k in NomicBertConfig.__dataclass_fields__})
    tower = BiEncoder(BiEncoderConfig(model_name="tiny", pooling="mean", logit_scale=20.0, trunk_config=tc), device=dev,
                      seed=3).train()
    scale = LogitScale(SimpleNamespace(logit_scale=20.0, trainable_logit_scale=False)).to(dev)
    return tower, scale


def _batch(G, S, vocab=512):
    g = torch.Generator().manual_seed(99)
    q = torch.randint(5, vocab, (G, S), generator=g)
    d = torch.randint(5, vocab, (G, S), generator=g)
    lens_q = torch.randint(S // 2, S + 1, (G,), generator=g).tolist()
    lens_d = torch.randint(S // 2, S + 1, (G,), generator=g).tolist()
    mq = (torch.arange(S)[None, :] < torch.tensor(lens_q)[:, None]).long()
    md = (torch.arange(S)[None, :] < torch.tensor(lens_d)[:, None]).long()
    return q, mq, d, md


def _step(rank, world, dev, G=16, S=32, chunk=4, overlap=True):
    from contrastors_amd.loss import grad_cache_loss

    tower, scale = _tower(dev)
    tower.overlap_reduce = overlap
    tower.broadcast_parameters(0)
    q, mq, d, md = _batch(G, S)
    b = G // world
    sl = slice(rank * b, (rank + 1) * b)
    qi = {"input_ids": q[sl].to(dev), "attention_mask": mq[sl].to(dev)}
    di = {"input_ids": d[sl].to(dev), "attention_mask": md[sl].to(dev)}
    tower.trunk.zero_grad()
    loss = grad_cache_loss(tower, qi, tower, di, chunk, scale)
    torch.cuda.synchronize()
    return float(loss), tower.trunk.flat_grad.detach().float().cpu().numpy()


def _worker(rank, world, port, out_dir, backend="gloo", exchange="rccl", resident="0", overlap=True):
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["CX_EXCHANGE"] = exchange
    os.environ["CX_GRADCACHE_RESIDENT"] = "auto" if resident == "tail" else resident
    if resident == "tail":   # the partially resident schedule (round 4): the last 4 of this rank's 8 document sequences + the
        import contrastors_amd.loss as L_   # last 6 query sequences keep their activations (a whole chunk + a shorter one)

        L_.resident_activations_fit = lambda *a, **k: False
        L_.resident_tail_plan = lambda *a, **k: (6, 4)
    local = rank if backend == "nccl" else 0   # RCCL: one rank per GPU; gloo: the ranks share the test box's one GPU
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    loss, grad = _step(rank, world, torch.device("cuda", local), overlap=overlap)
    used = 0
    from contrastors_amd import distributed as cxd

    if exchange in ("oneshot", "auto"):
        used = int(cxd._ONESHOT is not None and cxd._ONESHOT.epoch >= 2)   # one all-gather + one reduce-scatter at least
        if cxd._ONESHOT is not None:
            cxd._ONESHOT.check()
    np.savez(f"{out_dir}/w{rank}.npz", loss=loss, grad=grad, oneshot_used=used, report=json.dumps(cxd.exchange_report()))
    dist.barrier()
    dist.destroy_process_group()


def _oneshot_worker(rank, world, port, out_dir):
    """OneShotExchange on its own: repeated all-gathers / reduce-scatters of the metric's shard size against the values
    they must produce, a rank that runs ahead (no host sync between collectives), and the capacity check."""
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from contrastors_amd.distributed import OneShotExchange

    dev = torch.device("cuda", 0)
    n, d = 2048, 768
    ex = OneShotExchange(world * n * d * 4, device=dev)
    ok = True
    for it in range(6):
        mine = torch.full((n, d), float(10 * it + rank + 1), device=dev) + torch.arange(d, device=dev)[None] * 1e-3
        got = ex.all_gather(mine)
        want = torch.cat([torch.full((n, d), float(10 * it + r + 1), device=dev) + torch.arange(d, device=dev)[None] * 1e-3
                          for r in range(world)])
        ok &= bool(torch.equal(got, want))
        g = torch.stack([torch.full((n, d), float(100 * it + 10 * rank + p), device=dev) for p in range(world)]).view(world * n, d)
        rs = ex.reduce_scatter(g)
        want_rs = torch.full((n, d), float(sum(100 * it + 10 * r + rank for r in range(world))), device=dev)
        ok &= bool(torch.equal(rs, want_rs))
        if rank == 0 and it == 2:
            torch.cuda._sleep(200_000_000)   # rank 0 falls behind on the device: rank 1 must wait at the flags, not race ahead
    ex.check()
    too_big = False
    try:
        ex.all_gather(torch.zeros(2 * n + 8, d, device=dev))
    except ValueError:
        too_big = True
    torch.cuda.synchronize()
    np.savez(f"{out_dir}/x{rank}.npz", ok=int(ok), too_big=int(too_big), epochs=ex.epoch)
    ex.close()
    dist.barrier()
    dist.destroy_process_group()


def test_one_shot_exchange_two_ranks_sharing_the_gpu(tmp_path):
    """csrc/xgmi.hip + OneShotExchange: HIP-IPC receive buffers, peer stores, system-scope flags.  Two processes on the
    test box's one GPU exercise the whole protocol (the peer buffer is then reached through the same device, not through
    an xGMI link -- the semantics are the same, the bandwidth is for bench.py on a multi-GPU node to say)."""
    port = 29600 + (os.getpid() % 90)
    mp.spawn(_oneshot_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        x = np.load(tmp_path / f"x{r}.npz")
        assert int(x["ok"]) == 1 and int(x["too_big"]) == 1 and int(x["epochs"]) == 12


def test_two_rank_gradcache_step_with_the_one_shot_exchange(tmp_path):
    """gather_with_grad on the one-shot path (CX_EXCHANGE=oneshot) inside the real GradCache step: the same numbers as the
    process group's collectives."""
    port = 29500 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path), "gloo", "oneshot"), nprocs=2, join=True)
    a = [np.load(tmp_path / f"w{r}.npz") for r in range(2)]
    assert int(a[0]["oneshot_used"]) == 1 and int(a[1]["oneshot_used"]) == 1
    port += 1
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    mp.spawn(_worker, args=(2, port, str(ref_dir), "gloo", "rccl"), nprocs=2, join=True)
    b = [np.load(ref_dir / f"w{r}.npz") for r in range(2)]
    for r in range(2):
        assert float(a[r]["loss"]) == float(b[r]["loss"])
        denom = np.abs(b[r]["grad"]).max()
        assert np.abs(a[r]["grad"] - b[r]["grad"]).max() <= 1e-5 * denom   # (fp32 atomics of the LayerNorm reductions)


def test_two_rank_exchange_is_verified_and_chosen_by_measurement(tmp_path):
    """exchange = auto (the default): at the first gather_with_grad both ranks set the one-shot exchange
    up, check it bit-exact against the process group's collectives, time both and take the faster one -- the same verdict
    on every rank, recorded in exchange_report(); the step's numbers equal the process-group run's."""
    port = 29400 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path), "gloo", "auto"), nprocs=2, join=True)
    a = [np.load(tmp_path / f"w{r}.npz") for r in range(2)]
    reps = [json.loads(str(x["report"])) for x in a]
    for rep in reps:
        assert rep["mode"] == "auto" and rep["verified"] is True and rep["choice"] in ("oneshot", "pg"), rep
        assert rep["oneshot_us"] > 0 and rep["pg_us"] > 0 and rep["world"] == 2
        assert rep["choice"] == ("oneshot" if rep["oneshot_us"] < rep["pg_us"] else "pg")
    assert reps[0]["choice"] == reps[1]["choice"] and reps[0]["oneshot_us"] == reps[1]["oneshot_us"]   # MAX over ranks: identical
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    mp.spawn(_worker, args=(2, port + 1, str(ref_dir), "gloo", "rccl"), nprocs=2, join=True)
    b = [np.load(ref_dir / f"w{r}.npz") for r in range(2)]
    assert json.loads(str(b[0]["report"]))["choice"] == "pg"
    for r in range(2):
        assert float(a[r]["loss"]) == float(b[r]["loss"])
        assert np.abs(a[r]["grad"] - b[r]["grad"]).max() <= 1e-5 * np.abs(b[r]["grad"]).max()


@pytest.mark.parametrize("resident", ["0", "1", "tail"])
def test_overlapped_gradient_reduce_is_bit_identical_to_the_blocking_one(tmp_path, resident):
    """the step's last backward records one event per block (CxChunkBuffers.layer_events) and the flat
    gradient is all-reduced block by block on a side stream while the remaining blocks are still being differentiated;
    sync_gradients() only waits and rescales.  Same bits as one blocking all-reduce of the whole buffer, on both the
    two-pass, the resident and the partially resident ("tail": kept chunks are back-propagated first, the last re-forwarded
    chunk arms the reduction) GradCache schedules."""
    port = 29200 + (os.getpid() % 90) + (100 if resident == "1" else 200 if resident == "tail" else 0)
    ov, bl = tmp_path / "ov", tmp_path / "bl"
    ov.mkdir()
    bl.mkdir()
    mp.spawn(_worker, args=(2, port, str(ov), "gloo", "rccl", resident, True), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 1, str(bl), "gloo", "rccl", resident, False), nprocs=2, join=True)
    for r in range(2):
        a, b = np.load(ov / f"w{r}.npz"), np.load(bl / f"w{r}.npz")
        assert float(a["loss"]) == float(b["loss"])
        # the reduction adds the same two numbers either way; what two RUNS of the step differ by is the fp32-atomics order of
        # the type / position-row and bias reductions inside the backward (the same noise two blocking runs show)
        assert np.abs(a["grad"] - b["grad"]).max() <= 1e-5 * np.abs(b["grad"]).max()
        assert (a["grad"] == b["grad"]).mean() > 0.99
    # every rank ends the step with the SAME bits (a slice reduced twice or not at all would break this)
    np.testing.assert_array_equal(np.load(ov / "w0.npz")["grad"], np.load(ov / "w1.npz")["grad"])


def test_two_rank_gradcache_step_matches_single_process(tmp_path):
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    loss1, grad1 = _step(0, 1, torch.device("cuda", 0))
    w = [np.load(tmp_path / f"w{r}.npz") for r in range(2)]
    # identical reduced gradient on both ranks (one flat all-reduce)
    np.testing.assert_array_equal(w[0]["grad"], w[1]["grad"])
    # rank losses: CE over the rank's queries x W  ->  their mean is W x the global-batch loss
    assert abs((float(w[0]["loss"]) + float(w[1]["loss"])) / 2 - 2 * loss1) < 2e-3 * max(1.0, abs(loss1))
    g2, g1 = w[0]["grad"], 2.0 * grad1
    denom = np.abs(g1).max()
    assert denom > 0
    # same embeddings, same loss gradient; only the fp32 summation order / atomics and the bf16 rounding of the
    # per-rank activation gradients differ
    assert np.abs(g2 - g1).max() <= 2e-2 * denom
    assert np.abs(g2 - g1).mean() <= 2e-3 * denom


def test_bench_two_ranks_prints_one_json_line():
    env = dict(os.environ, CX_BENCH_BACKEND="gloo", CX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29800 + (os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup",
           "1", "--global-batch", "64", "--chunk-size", "16", "--layers", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["pairs_per_gpu"] == 32 and j["value"] > 0
    assert j["config"]["parallelism"] == "dp2" and np.isfinite(j["config"]["loss_last_step"])
    # the one-shot exchange record: set up, verified against the process group's all-gather and timed inside bench.py
    one = j["xgmi_allgather"]["oneshot"]
    assert isinstance(one, dict) and one["seconds"] > 0, one
    # ... by the data path itself, at its first exchange: verified against the process group, raced, the faster one taken
    ex = j["exchange"]
    assert ex["mode"] == "auto" and ex["verified"] is True and ex["choice"] in ("oneshot", "pg") and ex["oneshot_us"] > 0, ex
    assert j["xgmi_allgather"]["carried_by"] == ex["choice"]
    assert j["step_ms"]["n"] == 1 and j["step_ms"]["median"] > 0
    # round 5: the line is self-calibrating and answers the N > 1 questions by itself
    assert j["box"]["mfma_probe_tflops"] > 100 and j["box"]["hbm_copy_tbs"] > 0.5, j["box"]
    assert abs(j["frac_of_box_ceiling"] - j["roofline"]["achieved"] / j["box"]["mfma_probe_tflops"]) < 1e-9
    assert len(j["schedule_per_rank"]) == 2 and all(r["schedule"] in ("resident", "partial", "two-pass") for r in j["schedule_per_rank"])
    assert j["schedule_per_rank"][0]["schedule"] == j["schedule_per_rank"][1]["schedule"]   # the planner's budget is rank-agreed
    assert j["allreduce_exposed_ms"] >= 0 and j["allgather_us_rccl"] > 0 and 0 < j["allgather_xgmi_frac_rccl"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank (the 1-GPU test box skips this)")
def test_two_rank_gradcache_step_over_rccl(tmp_path):
    """The production collectives: all_gather_into_tensor / reduce_scatter_tensor of gather_with_grad
    (contrastors_amd/distributed.py, backend "nccl" = RCCL over xGMI) and the flat gradient all-reduce, one rank per GPU."""
    port = 29700 + (os.getpid() % 90)
    mp.spawn(_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    loss1, grad1 = _step(0, 1, torch.device("cuda", 0))
    w = [np.load(tmp_path / f"w{r}.npz") for r in range(2)]
    np.testing.assert_array_equal(w[0]["grad"], w[1]["grad"])
    assert abs((float(w[0]["loss"]) + float(w[1]["loss"])) / 2 - 2 * loss1) < 2e-3 * max(1.0, abs(loss1))
    g2, g1 = w[0]["grad"], 2.0 * grad1
    assert np.abs(g2 - g1).max() <= 2e-2 * np.abs(g1).max()


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no torch.distributed environment: bench.py spawns its own ranks
    and rank 0 prints exactly one JSON line with n_gpus = 2 (gloo on the shared test GPU; RCCL when launched for real)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CX_BENCH_BACKEND="gloo", CX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--global-batch", "64",
           "--chunk-size", "16", "--layers", "2", "--no-cpu-baseline", "--no-extra-legs"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["launch"].startswith("self") and j["value"] > 0


def test_two_rank_resident_gradcache_equals_two_pass(tmp_path):
    """CX_GRADCACHE_RESIDENT=1 on two ranks: pass 1 keeps the activations, the loss gathers across ranks in between, pass 2
    only back-propagates -- the same loss and (to fp32-atomics noise) the same reduced gradient as the two-pass schedule."""
    port = 29300 + (os.getpid() % 90)
    res_dir, ref_dir = tmp_path / "res", tmp_path / "ref"
    res_dir.mkdir()
    ref_dir.mkdir()
    mp.spawn(_worker, args=(2, port, str(res_dir), "gloo", "rccl", "1"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, port + 1, str(ref_dir), "gloo", "rccl", "0"), nprocs=2, join=True)
    for r in range(2):
        a, b = np.load(res_dir / f"w{r}.npz"), np.load(ref_dir / f"w{r}.npz")
        assert float(a["loss"]) == float(b["loss"])
        assert np.abs(a["grad"] - b["grad"]).max() <= 1e-5 * np.abs(b["grad"]).max()


def _metric_shape_worker(rank, world, port, out_dir):
    """Two tenants of one GPU, each with the PER-RANK shape of the metric's 8-GPU job (2048 pairs x 128 tokens of nomic-bert-2048,
    chunk 2048, train_args.gradcache_resident: auto -- the library default)."""
    sys.path.insert(0, str(ROOT))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["CX_EXCHANGE"] = "rccl"
    os.environ["CX_GRADCACHE_RESIDENT"] = "auto"   # (tests/conftest.py pins the suite to the two-pass schedule through the override)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from contrastors_amd import loss as L_
    from contrastors_amd.biencoder import BiEncoder, BiEncoderConfig, LogitScale
    from contrastors_amd.nomic_bert import NomicBertConfig
    from contrastors_amd.policy import GradCachePolicy

    tower = BiEncoder(BiEncoderConfig(model_name="nomic-ai/nomic-bert-2048", pooling="mean", logit_scale=50.0,
                                      trunk_config=NomicBertConfig.nomic_bert_2048()), device=dev, seed=0).train()
    tower.broadcast_parameters(0)
    scale = LogitScale(SimpleNamespace(logit_scale=50.0, trainable_logit_scale=False)).to(dev)
    g = torch.Generator().manual_seed(1234 + rank)
    b, S = 2048, 128
    q = {"input_ids": torch.randint(1000, 30522, (b, S), generator=g).to(dev), "seqlens": [S] * b}
    d = {"input_ids": torch.randint(1000, 30522, (b, S), generator=g).to(dev), "seqlens": [S] * b}
    pol = GradCachePolicy(chunk="exact", resident="auto")
    rec = []
    for step in range(2):
        tower.trunk.zero_grad()
        loss = L_.grad_cache_loss(tower, q, tower, d, 2048, scale, policy=pol)
        torch.cuda.synchronize()
        agreed = [v.get("free") for v in L_._AGREED_FREE.values()]
        rec.append(dict(L_.LAST_SCHEDULE, loss=float(loss), peak_gb=torch.cuda.max_memory_allocated(dev) / 1e9,
                        agreed_free_gb=(agreed[0] or 0) / 1e9 if agreed else -1.0, agreed_keys=[str(k) for k in L_._AGREED_FREE]))
    tower.trunk.drop_idle_arenas()            # (the device is shared with another tenant: give the pooled arenas back first)
    gsum = float(tower.trunk.flat_grad.norm())   # (a reduction without a device-sized temporary)
    json.dump({"steps": rec, "grad_abs_sum": gsum}, open(f"{out_dir}/m{rank}.json", "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.environ.get("CX_TEST_TWO_TENANTS"),
                    reason="stress test: two tenants fill one 288 GB device on purpose (each wants 171 GB); it needs the device to itself -- a pytest "
                           "process that has run the rest of the suite is a third tenant.  Run it on its own: CX_TEST_TWO_TENANTS=1 pytest "
                           "tests/test_distributed_gpu.py::test_two_tenants_at_the_metric_per_rank_shape (profiles/r5_two_tenants_planner.txt)")
def test_two_tenants_at_the_metric_per_rank_shape(tmp_path):
    """the memory planners with two tenants on one device, each at the 8-GPU job's per-rank shape.  Each rank
    alone would keep everything resident (171 GB of 288); two of them cannot.  What must hold: the planners' budget is the ranks'
    agreed minimum (same schedule decision on both), whoever runs out of memory falls back without leaving a collective
    unmatched (no hang: both ranks finish), the losses are finite and the reduced gradients identical on both ranks."""
    port = 29100 + (os.getpid() % 90)
    mp.spawn(_metric_shape_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [json.load(open(tmp_path / f"m{k}.json")) for k in range(2)]
    print("two tenants:", json.dumps(r))
    for k in range(2):
        assert all(np.isfinite(s["loss"]) for s in r[k]["steps"]), r[k]
    assert abs(r[0]["grad_abs_sum"] - r[1]["grad_abs_sum"]) <= 1e-6 * r[0]["grad_abs_sum"]
    # step 0: the figure the ranks agreed on -- except that a rank which ran out of memory has clamped ITS copy to what it measured then
    # (loss.note_local_memory_shortfall, round 6: no collective inside a fallback), so it may only be smaller; step 1 starts with the
    # second agreement of the process group: identical again on both ranks
    a0 = [r[k]["steps"][0]["agreed_free_gb"] for k in range(2)]
    fb0 = [bool(r[k]["steps"][0]["fell_back"]) for k in range(2)]
    if not any(fb0):
        assert a0[0] == a0[1] > 0, r
    else:
        assert all(0 < a0[k] <= max(a0) for k in range(2)) and all(a0[k] == max(a0) for k in range(2) if not fb0[k]), r
    assert r[0]["steps"][1]["agreed_free_gb"] == r[1]["steps"][1]["agreed_free_gb"] > 0, r
    assert [s["schedule"] for s in r[0]["steps"]] == [s["schedule"] for s in r[1]["steps"]] or any(s["fell_back"] for k in range(2) for s in r[k]["steps"]), r
