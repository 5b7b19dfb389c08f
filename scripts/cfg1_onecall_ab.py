"""Same-process A/B of BASELINE configs[0] (bench.py leg_cfg1): one tower call per direct step (trainers.encode_pair) vs the
reference's two calls.  usage: python scripts/cfg1_onecall_ab.py [--steps 30]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

import bench  # noqa: E402
from contrastors_amd import trainers as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
limit = T.PAIR_FUSE_MAX_TOKENS
print("# cfg 1 (bert-base-uncased, B = 32, S = 64, direct step), pairs/s; alternating in one process")
for rnd in range(3):
    row = []
    for name, lim in (("two calls", 0), ("one call", limit)):
        T.PAIR_FUSE_MAX_TOKENS = lim
        for drop in (True, False):
            r = bench.leg_cfg1(torch, dev, a.steps, hf_dropout=drop)
            row.append(f"{name} dropout {0.1 if drop else 0.0}: {r['value']:7.1f} ({r['ms_per_step']:.2f} ms)")
    print(" | ".join(row))
T.PAIR_FUSE_MAX_TOKENS = limit
