"""Batch / wire contract of the contrastive streaming loader: host-side mirror of
`StreamingShardDataset` (sc/dataset/text_text_loader.py:150-672) for shards on a local / mounted file system.
The S3 / R2 transport (fsspec pipes, download-to-/tmp, webdataset URL helpers) is NOT rebuilt; everything a trainer
or a data producer has to agree on is:

* spec YAML (`datasets: [{name, bucket, objective: {type, columns}, weight?, kd_loss?, query_only?, query_prefix?,
  document_prefix?}]`, :262-350), brace-range shard lists, `counts.json` / `offsets.json.gz` next to the shards
  (README.md:102-121; keys = normalised URLs :250-260, offsets are [start, end) byte ranges of the DECOMPRESSED stream),
* which records a rank reads: `max_per_shard = floor(count / world / rank_batch) * rank_batch` (:310-318), shard choice
  by `random.Random(seed)` shared by all ranks (:468-485), record index `processed * world + rank * rank_batch`
  (:482-510), per-rank progress file `rank_{rank}_processed_{run_name}.json` and resume from it (:352-380),
* record -> pair mapping incl. negative folding `document = [positive] + negatives[:N]` flattened in record order
  (:575-603), knowledge-distillation scores (:605-613),
* text -> tensors: optional EOS suffix, prefix rules (`query: ` / `passage: `, per-dataset override, query-only
  datasets :632-644), `padding="max_length"`, truncation to 32 / 256 tokens unless overridden (:23,197-200,646), last
  token forced to EOS (:650-651), keys `{query,document}_{input_ids,attention_mask,...}`, `dataset_name`, `kd_scores`.

tests/golden/loader_*.npz hold what the reference class yields for the same shards on 2 ranks.
"""
from __future__ import annotations

import gzip
import json
import os
import random
import re
from pathlib import Path
from typing import Dict, Iterator, List, Optional

import torch
import torch.distributed as dist
import yaml
from torch.utils.data import IterableDataset

MAPPED_NAMES = {"paired": ["query", "document"], "self": ["query"], "triplet": ["query", "document", "negative"]}
KEY2PREFIX = {"query": "query", "document": "passage", "negative": "passage"}
DEFAULT_COL_TO_MAX_TOKENS = {"query": 32, "document": 256, "negative": 256}
_SPEC_KEYS = set("name bucket objective weight kd_loss query_only query_prefix document_prefix".split())
_BRACE = re.compile(r"\{([^{}]*)\}")


def expand_urls(pattern: str) -> List[str]:
    """Brace expansion of shard lists: `{00000..00538}` ranges (zero padding kept) and `{a,b}` alternatives."""
    m = _BRACE.search(pattern)
    if not m:
        return [pattern]
    body, out = m.group(1), []
    rng = re.fullmatch(r"(-?\d+)\.\.(-?\d+)", body)
    if rng:
        a, b = rng.group(1), rng.group(2)
        width = max(len(a), len(b)) if (a.startswith("0") or b.startswith("0")) and len(a) == len(b) else 0
        lo, hi = int(a), int(b)
        step = 1 if hi >= lo else -1
        items = [str(i).zfill(width) for i in range(lo, hi + step, step)]
    else:
        items = body.split(",")
    for it in items:
        out.extend(expand_urls(pattern[: m.start()] + it + pattern[m.end():]))
    return out


def normalize_url(url: str) -> str:
    """The key under which counts.json / offsets.json.gz index a shard: the last 3 path components, or the last 4 when
    the URL has 6 or more `/`-separated parts (sc/dataset/text_text_loader.py:250-260)."""
    parts = url.split("/")
    return "/".join(parts[-4:] if len(parts) >= 6 else parts[-3:])


def build_index(shard_paths: List[str], out_dir: Optional[str] = None) -> Dict[str, int]:
    """Write `counts.json` and `offsets.json.gz` for gzip'ed jsonl shards (the producer side of README.md:102-121)."""
    counts, offsets = {}, {}
    for p in shard_paths:
        key, per, pos = normalize_url(p), {}, 0
        with gzip.open(p, "rb") as f:
            for i, line in enumerate(f):
                per[str(i)] = [pos, pos + len(line)]
                pos += len(line)
        counts[key], offsets[key] = len(per), per
    out = Path(out_dir or Path(shard_paths[0]).parent)
    with open(out / "counts.json", "w") as f:
        json.dump({"count_per_file": counts}, f)
    with gzip.open(out / "offsets.json.gz", "wt") as f:
        json.dump(offsets, f)
    return counts


def collate_fn(batch):
    """The dataset yields whole per-rank batches; the DataLoader runs with batch_size=1 (text_text.py:213)."""
    return batch[0]


class StreamingShardDataset(IterableDataset):
    def __init__(self, ds_spec: str, global_batch_size: int, tokenizer, seed: int, add_eos: bool = True,
                 add_prefix: bool = False, num_negatives: int = -1, download_locally: bool = False,
                 process_one_shard: bool = False, weighted_sampling: bool = False, verbose: bool = True,
                 infinite: bool = False, sample_negatives: bool = False, run_name: Optional[str] = None,
                 query_max_length: Optional[int] = None, document_max_length: Optional[int] = None,
                 state_dir: Optional[str] = None):
        if download_locally:
            raise NotImplementedError("remote shards (S3 / R2 download) are outside the built path: mount them locally")
        self.global_batch_size, self.tokenizer = global_batch_size, tokenizer
        self.rng = random.Random(seed)
        self.add_eos, self.add_prefix, self.num_negatives = add_eos, add_prefix, num_negatives
        self.process_one_shard, self.weighted_sampling = process_one_shard, weighted_sampling
        self.verbose, self.infinite, self.sample_negatives, self.run_name = verbose, infinite, sample_negatives, run_name
        self.current_shard = None
        if query_max_length is not None and document_max_length is not None:
            self.col_max_length = {"query": query_max_length, "document": document_max_length,
                                   "negative": document_max_length}
        else:
            self.col_max_length = dict(DEFAULT_COL_TO_MAX_TOKENS)
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        else:
            self.rank, self.world_size = 0, 1
        if global_batch_size % self.world_size:
            raise ValueError(f"global batch {global_batch_size} is not divisible by {self.world_size} ranks")
        self.rank_batch_size = global_batch_size // self.world_size
        self.num_samples_per_shard: Dict[str, int] = {}
        self.max_per_shard: Dict[str, int] = {}
        self.max_per_ds: Dict[str, int] = {}
        self.path2objective: Dict[str, dict] = {}
        self.path2offsets: Dict[str, dict] = {}
        self.path2prefix: Dict[str, Dict[str, str]] = {}
        self.kd_loss: Dict[str, bool] = {}
        self.query_only = set()
        self.total_samples = 0
        self.ds_paths = self.parse_spec(ds_spec)
        self.current_paths = list(self.ds_paths)
        self._streams: Dict[str, gzip.GzipFile] = {}
        state_dir = state_dir or ds_spec.replace(".yaml", "")
        os.makedirs(state_dir, exist_ok=True)
        self.path = f"{state_dir}/rank_{self.rank}_processed_{self.run_name}.json"
        self._processed = {p: 0 for p in self.ds_paths}
        self._persist()
        if self.weighted_sampling:
            self.weights = self.calculate_weights()

    # ---- spec ----------------------------------------------------------------------------------------------------------
    def parse_spec(self, fname: str) -> List[str]:
        with open(fname) as f:
            spec = yaml.safe_load(f)
        paths: List[str] = []
        for ds in spec["datasets"]:
            if not set(ds.keys()) <= _SPEC_KEYS:
                raise AssertionError(list(ds.keys()))
            urls = expand_urls(ds["bucket"])
            if any(u.startswith("s3://") for u in urls):
                raise NotImplementedError("s3:// shards: mount the bucket and point `bucket` at the local path")
            bucket = "/".join(ds["bucket"].split("/")[:-1])
            with open(f"{bucket}/counts.json") as f:
                counts = json.load(f)
            counts = counts.get("count_per_file", counts)
            with gzip.open(f"{bucket}/offsets.json.gz", "rt") as f:
                self.path2offsets[bucket] = json.load(f)
            keep = []
            per_rank_total = 0
            for url in urls:
                key = normalize_url(url)
                n = counts.get(key, 0)
                per_rank = int(n / self.world_size / self.rank_batch_size) * self.rank_batch_size
                if per_rank == 0:
                    continue  # a shard too small for one global batch contributes nothing
                keep.append(url)
                self.path2objective[key] = ds["objective"]
                self.num_samples_per_shard[key] = n
                self.max_per_shard[key] = per_rank
                self.kd_loss[url] = ds.get("kd_loss", False)
                per_rank_total += per_rank
            paths.extend(keep)
            self.max_per_ds[ds["name"]] = per_rank_total * self.world_size
            self.total_samples += per_rank_total * self.world_size
            ds_name = Path(ds["bucket"]).parent.name
            if ds.get("query_only", False):
                self.query_only.add(ds_name)
            if ds.get("query_prefix"):
                doc = ds.get("document_prefix", ds["query_prefix"])
                self.path2prefix[ds_name] = {"query": ds["query_prefix"], "document": doc}
                if self.num_negatives > 0:
                    self.path2prefix[ds_name]["negative"] = doc
        return paths

    def __len__(self):
        return self.total_samples

    # ---- progress ------------------------------------------------------------------------------------------------------
    def _persist(self):
        with open(self.path, "w") as f:
            json.dump(self._processed, f, indent=3)

    def state_dict(self) -> Dict[str, int]:
        return dict(self._processed)

    def save_state(self, output_dir: str):
        """What the trainer copies into a checkpoint (`rank_{rank}_processed.json`, text_text_loader.py:358)."""
        os.makedirs(output_dir, exist_ok=True)
        with open(f"{output_dir}/rank_{self.rank}_processed.json", "w") as f:
            json.dump(self._processed, f, indent=3)

    def load_state(self, path: str):
        with open(f"{path}/rank_{self.rank}_processed.json") as f:
            self._processed = {k: int(v) for k, v in json.load(f).items()}
        self._persist()
        self.current_paths = [p for p in self.ds_paths if self._processed[p] < self.max_per_shard[normalize_url(p)]]

    def calculate_weights(self) -> Dict[str, float]:
        total = sum(self.num_samples_per_shard.values())
        return {p: (self.num_samples_per_shard[normalize_url(p)] - self._processed[p] * self.world_size) / total
                for p in self.ds_paths}

    # ---- reading -------------------------------------------------------------------------------------------------------
    def _next_shard(self) -> str:
        if self.process_one_shard:
            if self.current_shard is None:
                self.current_shard = self.rng.choice(self.current_paths)
            return self.current_shard
        if self.weighted_sampling:
            w = [self.weights[p] for p in self.current_paths]
            return self.rng.choices(self.current_paths, weights=w, k=1)[0]
        return self.rng.choice(self.current_paths)

    def _read_records(self, path: str) -> List[dict]:
        key = normalize_url(path)
        offsets = self.path2offsets["/".join(path.split("/")[:-1])][key]
        first = self._processed[path] * self.world_size + self.rank * self.rank_batch_size
        stream = self._streams.get(path)
        if stream is None:
            stream = self._streams[path] = gzip.open(path, "rb")
        if stream.tell() != offsets[str(first)][0]:
            stream.seek(offsets[str(first)][0])
        objective = self.path2objective[key]
        out = []
        for i in range(first, min(first + self.rank_batch_size, len(offsets))):
            start, end = offsets[str(i)]
            data = json.loads(stream.read(end - start).decode())
            sample = self.extract_pair(data, objective, path)
            if self.kd_loss[path]:
                sample["kd_scores"] = [data["document_score"]] + data["negatives_scores"][: self.num_negatives]
            out.append(sample)
        return out

    def extract_pair(self, data: dict, objective: dict, path: str) -> dict:
        ctype, columns = objective["type"], objective["columns"]
        valid = data["metadata"]["objective"][ctype]
        if columns not in valid:
            raise AssertionError(f"Invalid columns {columns} for contrastive type {ctype}. Valid columns are {valid}")
        pair = {}
        for mapped, col in zip(MAPPED_NAMES[ctype], columns):
            if mapped != "negative":
                pair[mapped] = data[col]
                continue
            negs = data[col]
            if len(negs) > self.num_negatives:
                negs = random.sample(negs, self.num_negatives) if self.sample_negatives else negs[: self.num_negatives]
            pair["document"] = [pair["document"]] + list(negs)
        base = re.match(r"^((?:.*/|)[^.]+)[.]([^/]*)$", path)  # webdataset's base_plus_ext
        pair["__key__"] = f"{base.group(1)}.{base.group(2).lower()}"
        return pair

    def __iter__(self) -> Iterator[dict]:
        while True:
            while self.current_paths:
                path = self._next_shard()
                key = normalize_url(path)
                batch = self._read_records(path)
                self._processed[path] += len(batch)
                self._persist()
                if self._processed[path] >= self.max_per_shard[key]:
                    self.current_paths.remove(path)
                    self._streams.pop(path).close()
                    if self.process_one_shard:
                        self.current_shard = None
                if len(batch) < self.rank_batch_size:
                    raise ValueError(f"Batch size {len(batch)} is too small, something went wrong on rank {self.rank} "
                                     f"for path {path}")
                yield self.tokenize_pairs(batch, self.path2objective[key])
                if self.weighted_sampling:
                    self.weights = self.calculate_weights()
            if not self.infinite:
                break
            self.current_paths = list(self.ds_paths)
            self._processed = {p: 0 for p in self.ds_paths}
            self._persist()

    # ---- text -> tensors -----------------------------------------------------------------------------------------------
    def tokenize_pairs(self, samples: List[dict], objective: dict) -> dict:
        key = samples[0]["__key__"]
        dataset_name = key.split("/")[-2]
        if "mc4" in key:
            dataset_name = f"mc4_{dataset_name}"
        elif "multilingual-cc-news" in key:
            dataset_name = f"cc_news_{dataset_name}"
        out = {"dataset_name": dataset_name}
        eos = self.tokenizer.eos_token if self.add_eos else ""
        for col in MAPPED_NAMES[objective["type"]]:
            if col == "negative":
                continue  # already folded into `document`
            texts: List[str] = []
            for s in samples:
                v = s[col]
                texts.extend([t + eos for t in v] if isinstance(v, list) else [v + eos])
            if self.add_prefix and not (dataset_name in self.query_only and col != "query"):
                if dataset_name in self.path2prefix:
                    prefix = self.path2prefix[dataset_name][col]
                elif dataset_name in self.query_only:
                    prefix = "query"
                else:
                    prefix = KEY2PREFIX[col]
                texts = [f"{prefix}: {t}" for t in texts]
            tok = self.tokenizer(texts, padding="max_length", truncation=True, return_tensors="pt",
                                 max_length=self.col_max_length[col])
            if self.add_eos:
                tok["input_ids"][:, -1] = self.tokenizer.eos_token_id
            out.update({f"{col}_{k}": v for k, v in tok.items()})
        if "kd_scores" in samples[0]:
            out["kd_scores"] = torch.tensor([s["kd_scores"] for s in samples], dtype=torch.float32)
        return out


class LocalShardDataset(torch.utils.data.Dataset):
    """Map-style twin for small corpora (sc/dataset/text_text_loader.py:660-760): every record of every shard is read
    into memory; `__getitem__` folds `num_negatives` sampled negatives into `document` the first time a record is
    touched (the reference pops `negative` from the stored record, so later epochs see the same sample -- kept)."""

    def __init__(self, ds_spec: str, num_negatives: int = 0, seed: int = 42):
        if dist.is_available() and dist.is_initialized():
            self.rank, self.world_size = dist.get_rank(), dist.get_world_size()
        else:
            self.rank, self.world_size = 0, 1
        self.path2objective: Dict[str, dict] = {}
        self.path2prefix: Dict[str, Dict[str, str]] = {}
        self.query_only = set()
        self.num_negatives = num_negatives
        self.rng = random.Random(seed)
        self.local_paths = self.parse_spec(ds_spec)
        self.examples = self.load_examples(self.local_paths)

    def parse_spec(self, fname: str) -> List[str]:
        with open(fname) as f:
            spec = yaml.safe_load(f)
        paths: List[str] = []
        for ds in spec["datasets"]:
            if not set(ds.keys()) <= _SPEC_KEYS:
                raise AssertionError(list(ds.keys()))
            urls = expand_urls(ds["bucket"])
            if any(u.startswith("s3://") for u in urls):
                raise NotImplementedError("s3:// shards: mount the bucket and point `bucket` at the local path")
            paths.extend(urls)
            self.path2objective.update({u: ds["objective"] for u in urls})
            ds_name = Path(ds["bucket"]).parent.name
            if ds.get("query_only", False):
                self.query_only.add(ds_name)
            if ds.get("query_prefix"):
                doc = ds.get("document_prefix", ds["query_prefix"])
                self.path2prefix[ds_name] = {"query": ds["query_prefix"], "document": doc}
                if self.num_negatives > 0:
                    self.path2prefix[ds_name]["negative"] = doc
        return paths

    def load_examples(self, paths: List[str]) -> List[dict]:
        examples = []
        for path in paths:
            ds_name = Path(path).parent.name
            objective = self.path2objective[path]
            ctype, columns = objective["type"], objective["columns"]
            opener = gzip.open if str(path).endswith(".gz") else open
            with opener(path, "rt") as f:
                for line in f:
                    if not line.strip():
                        continue
                    rec = json.loads(line)
                    valid = rec["metadata"]["objective"][ctype]
                    if columns not in valid:
                        raise AssertionError(f"Invalid columns {columns} for contrastive type {ctype}. Valid columns "
                                             f"are {valid}")
                    mapped = {name: rec[col] for name, col in zip(MAPPED_NAMES[ctype], columns)}
                    mapped["dataset_name"] = ds_name
                    examples.append(mapped)
        return examples

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, index):
        data = self.examples[index]
        if "negative" in data:
            negatives = data.pop("negative")
            data["document"] = [data["document"]] + random.sample(negatives, self.num_negatives)
        return data


def collate_local_ds(batch, tokenizer, add_prefix: bool = False, query_only=None, path2prefix=None) -> dict:
    """sc/dataset/text_text_loader.py:763-800: one tokenizer call per column (`padding="max_length"`, the tokenizer's
    own maximum), list columns flattened in record order, prefix chosen per record by its dataset."""
    batch = [dict(sample) for sample in batch]  # the reference pops `dataset_name` from the stored records themselves;
    ds_names = [sample.pop("dataset_name") for sample in batch]  # a copy keeps later epochs usable
    out = {}
    for col in batch[0].keys():
        collected = [sample[col] for sample in batch]
        if isinstance(collected[0], list):
            ds_names = sum([[n] * len(sample[col]) for n, sample in zip(ds_names, batch)], [])
            collected = sum(collected, [])
        if add_prefix:
            if path2prefix:
                prefixes = [path2prefix[n][col] for n in ds_names]
            elif query_only and col != "query":
                prefixes = ["query" if n in query_only else KEY2PREFIX[col] for n in ds_names]
            else:
                prefixes = [KEY2PREFIX[col]] * len(collected)
            collected = [f"{pre}: {text}" for pre, text in zip(prefixes, collected)]
        tok = tokenizer(collected, padding="max_length", truncation=True, return_tensors="pt")
        out.update({f"{col}_{k}": v for k, v in tok.items()})
    return out


def get_local_dataloader(ds_spec: str, batch_size: int, tokenizer, num_negatives: int, seed: int, add_prefix: bool,
                         num_workers: int = 0, epoch: int = 0):
    """sc/dataset/text_text_loader.py:803-823: per-rank batches through a seeded DistributedSampler, drop_last."""
    from torch.utils.data import DataLoader, DistributedSampler

    dataset = LocalShardDataset(ds_spec, num_negatives=num_negatives, seed=seed)
    sampler = None
    if dist.is_available() and dist.is_initialized():
        sampler = DistributedSampler(dataset, shuffle=True, num_replicas=dist.get_world_size(), rank=dist.get_rank(),
                                     seed=seed)
        sampler.set_epoch(epoch)

    def collate(x):
        return collate_local_ds(x, tokenizer, add_prefix=add_prefix, query_only=dataset.query_only,
                                path2prefix=dataset.path2prefix)

    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, collate_fn=collate, drop_last=True,
                      num_workers=num_workers)


def get_streaming_dataset(config, tokenizer, run_name: Optional[str] = None) -> StreamingShardDataset:
    """The construction sc/trainers/text_text.py:191-209 does from the YAML sections."""
    da, ma = config.data_args, config.model_args
    return StreamingShardDataset(
        da.input_shards, da.batch_size, tokenizer, seed=da.seed, add_eos=ma.nomic_encoder is not True,
        add_prefix=ma.add_prefix, num_negatives=ma.num_negatives, download_locally=bool(da.download),
        process_one_shard=bool(da.process_one_shard), weighted_sampling=bool(da.weighted_sampling),
        verbose=bool(da.verbose), sample_negatives=bool(da.sample_negatives), run_name=run_name,
        query_max_length=da.query_max_length, document_max_length=da.document_max_length)
