"""`python -m contrastors_amd.train --config X.yaml --dtype bf16 [--key value ...]` (mirror of sc/train.py:51-131).

torchrun / `python -m torch.distributed.run` launches one process per GPU; backend "nccl" is RCCL on ROCm.  CLI flags
named like fields of train_args / model_args / data_args override the YAML (sc/train.py:87-94).  Data: with
`data_args.input_shards` pointing at a spec YAML of LOCAL shards (contrastors_amd/data.py; S3 transport is not built) and
`--synthetic-steps 0` the streaming loader feeds the trainer (the tokenizer named by `model_args.tokenizer_name` must be
available offline); otherwise `--synthetic-steps N` drives it with synthetic batches of the loader's exact key contract.
"""
from __future__ import annotations

import argparse
import os

import torch
import torch.distributed as dist

from .config import read_config
from .trainers import TRAINER_REGISTRY, synthetic_batches


def _str2bool(v):
    """sc/train.py:39-47."""
    if isinstance(v, bool):
        return v
    t = str(v).lower()
    if t in ("yes", "true", "t", "y", "1"):
        return True
    if t in ("no", "false", "f", "n", "0"):
        return False
    raise ValueError(f"Boolean value expected, got {v!r}")


def split_overrides(extra):
    """`--key value`, `--key=value` and a bare `--flag` (= true, as the reference's `nargs='?', const=True` flags) -> dict."""
    out, i = {}, 0
    while i < len(extra):
        tok = extra[i]
        if not tok.startswith("--"):
            raise SystemExit(f"unexpected argument {tok!r}")
        key = tok[2:]
        if "=" in key:
            key, val = key.split("=", 1)
        elif i + 1 < len(extra) and not extra[i + 1].startswith("--"):
            val = extra[i + 1]
            i += 1
        else:
            val = True
        out[key.replace("-", "_")] = val
        i += 1
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "bfloat16"])   # sc/train.py:29-36: the names of bf16 there
    ap.add_argument("--local_rank", type=int, default=-1)                      # sc/train.py:54 (launchers that still pass it)
    ap.add_argument("--synthetic-steps", type=int, default=10)
    ap.add_argument("--seq-len", type=int, default=128)
    args, extra = ap.parse_known_args(argv)
    return args, split_overrides(extra)


def apply_overrides(config, overrides):
    """sc/train.py:87-94 `update_config_with_args`: a key is written into every section that has it.  The reference declares a
    fixed list of flags; here any field of the three sections can be overridden, typed by its current value / annotation --
    and a key no section has is an error, not a silent no-op."""
    sections = [s for s in (config.train_args, config.model_args, config.data_args) if s is not None]
    for k, v in overrides.items():
        hit = False
        for section in sections:
            if k not in type(section).model_fields:
                continue
            hit = True
            cur = getattr(section, k)
            ann = str(type(section).model_fields[k].annotation)
            if isinstance(cur, bool) or (cur is None and "bool" in ann):
                val = _str2bool(v)
            elif isinstance(cur, int) and not isinstance(v, bool):
                val = int(v)
            elif isinstance(cur, float) or (cur is None and "float" in ann and "str" not in ann):
                val = float(v)
            elif cur is None and "int" in ann and "str" not in ann:
                val = int(v)
            else:
                val = v
            setattr(section, k, val)
        if not hit:
            raise SystemExit(f"--{k}: no such key in train_args / model_args / data_args")
    return config


def main():
    args, overrides = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    config = apply_overrides(read_config(args.config), overrides)
    if config.model_args.model_type not in TRAINER_REGISTRY:   # glue / mmlm / distill: valid reference types, not on this path
        raise NotImplementedError(f"model_type {config.model_args.model_type!r}: this build serves {sorted(TRAINER_REGISTRY)} "
                                  "(the contrastive, image-text and MLM trainers of the hot path)")
    trainer = TRAINER_REGISTRY[config.model_args.model_type](config, torch.bfloat16, total_steps=args.synthetic_steps)
    per_rank = config.data_args.batch_size // world
    if args.synthetic_steps <= 0 and config.data_args.input_shards:
        from transformers import AutoTokenizer

        from .data import get_streaming_dataset

        tok = AutoTokenizer.from_pretrained(config.model_args.tokenizer_name, local_files_only=True)
        if config.data_args.streaming:
            ds = get_streaming_dataset(config, tok, run_name=getattr(config.train_args, "wandb_run_name", None) or "run")
            trainer.set_total_steps(len(ds) // config.data_args.batch_size)  # before the first step: the LR horizon
            trainer.train(iter(ds), log_every=10)
        else:  # sc/trainers/text_text.py:228-244: map-style dataset + DistributedSampler, per-rank batch
            from .data import get_local_dataloader

            dl = get_local_dataloader(config.data_args.input_shards, per_rank, tok, seed=config.data_args.seed,
                                      num_negatives=config.model_args.num_negatives,
                                      add_prefix=config.model_args.add_prefix, num_workers=config.data_args.workers)
            trainer.set_total_steps(len(dl.dataset) // config.data_args.batch_size)
            trainer.train(iter(dl), log_every=10)
    elif config.model_args.model_type == "mlm":
        from .mlm import synthetic_mlm_batches

        prob = config.data_args.mlm_prob if config.data_args.mlm_prob is not None else 0.3  # mlm.yaml: 0.30
        rank = dist.get_rank() if world > 1 else 0
        trainer.train(synthetic_mlm_batches(args.synthetic_steps, per_rank, args.seq_len, mlm_probability=prob,
                                            seed=1234 + rank), log_every=1)
    else:
        trainer.train(synthetic_batches(args.synthetic_steps, per_rank, args.seq_len, rank=trainer.rank), log_every=1)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
