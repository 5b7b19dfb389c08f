"""RNG snapshot / replay for GradCache re-forwards (role of sc/rand_state.py:6-22 `RandContext`).

`grad_cache_loss` takes one snapshot per chunk before the no-grad forward of pass 1 and replays it around the re-forward
of pass 2 (sc/loss.py:141-145,156-158).  The native engine draws the Philox (seed, offset) of its dropout masks from
torch's device generator (NomicBertEngine._arm_dropout), so restoring that generator regenerates the masks bit for bit.
With dropout 0 (every BASELINE config) no random number is consumed and the snapshot is skipped (`needed=False`), which
also avoids the per-chunk device sync the reference pays.  The context captures the CPU generator
and the generator of every CUDA/HIP device that owns one of the chunk's tensors at construction, and replays them inside
`with ctx:`, restoring the outer state on exit.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch


def _devices_of(tensors: Iterable) -> list:
    seen = []
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda and t.device.index not in seen:
            seen.append(t.device.index)
    return seen


class RandContext:
    def __init__(self, tensors, needed: bool = True):
        items = tensors.values() if isinstance(tensors, dict) else tensors
        self.needed = needed
        self._cpu: Optional[torch.Tensor] = None
        self._dev: Dict[int, torch.Tensor] = {}
        self._outer_cpu = None
        self._outer_dev: Dict[int, torch.Tensor] = {}
        if needed:
            self._cpu = torch.get_rng_state()
            for idx in _devices_of(items):
                self._dev[idx] = torch.cuda.get_rng_state(idx)

    def restore(self):
        """Set the generators back to the snapshot for good (no outer state is kept): a forward that failed half-way may
        have advanced them already, and its retry has to start where pass 2's replay will start."""
        if self.needed:
            torch.set_rng_state(self._cpu)
            for i, s in self._dev.items():
                torch.cuda.set_rng_state(s, i)

    def __enter__(self):
        if not self.needed:
            return self
        self._outer_cpu = torch.get_rng_state()
        self._outer_dev = {i: torch.cuda.get_rng_state(i) for i in self._dev}
        torch.set_rng_state(self._cpu)
        for i, s in self._dev.items():
            torch.cuda.set_rng_state(s, i)
        return self

    def __exit__(self, *exc):
        if self.needed:
            torch.set_rng_state(self._outer_cpu)
            for i, s in self._outer_dev.items():
                torch.cuda.set_rng_state(s, i)
        return False
