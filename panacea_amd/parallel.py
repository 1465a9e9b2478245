"""Multi-GPU execution of the denoising path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X, "gloo" in the CPU tests).

What shards, and how:

* **samples** — the reference's own strategy (`inference.py:248-280`: DistributedSampler + a DDP wrapper used only
  for `.module`): independent units, no collective on the data path.  `replica_seed()` mirrors `inference.py:250`.
* **CFG halves** — inside `OpenAIWrapperControlLDM3D.forward` the unconditional and conditional halves of the
  guidance batch never interact (every op is per sample; GroupNorm statistics are per frame).  Only
  `VanillaCFG.__call__` (guiders.py:25-29) combines them.  `ShardedCFG` is a drop-in guider that gives each rank
  of a pair ONE half (half the frames per network evaluation => half the latency per step) and exchanges the
  denoised halves with one all-gather of (T, 4, h, w) per step — 0.8 MB at the nuScenes shape, the only real
  exchange step of the path; each peer pair has its own xGMI link, so pairs do not contend.
* **frame groups / view groups** inside a half are NOT sharded here: every ResBlock3D (temporal GroupNorm +
  conv1d, 64 sites/step) and every temporal attention (23 sites/step) needs all T frames of a pixel, i.e. an
  all-gather of a full-resolution activation per site (≈ 94 MB received per L0 site at 4 frame groups, ≈ 3-4 GB
  per step and rank — as much time on 7 x 153 GB/s links as the compute it would save), and every 3x3 conv and
  spatial GroupNorm couples the views.  Recorded as the next row in DESIGN.md §9.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .sampling import VanillaCFG


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from the torchrun environment; returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def replica_seed(rank: int, base: int = 3407) -> int:
    """inference.py:250 — every replica draws its own noise."""
    return base + rank


def cfg_pair_groups(world: int) -> List[Optional[dist.ProcessGroup]]:
    """One 2-rank group per sample: ranks (2k, 2k+1).  Every rank must call this (collective group creation)."""
    if world % 2:
        raise ValueError("CFG sharding needs an even number of ranks")
    return [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]


class ShardedCFG(VanillaCFG):
    """VanillaCFG over a pair of ranks: rank 2k evaluates the unconditional half, rank 2k+1 the conditional
    half (the reference's batch order, uncond first: guiders.py:36,40)."""

    def __init__(self, scale: float, group: dist.ProcessGroup, half: int):
        super().__init__(scale)
        self.group, self.half = group, half

    def prepare_inputs(self, x, s, c, uc):
        src = uc if self.half == 0 else c
        c_out: Dict = {}
        for k in c:
            if k in self.KEYS:
                c_out[k] = src[k]
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return x, s, c_out

    def __call__(self, x, sigma):
        parts = [torch.empty_like(x), torch.empty_like(x)]
        dist.all_gather(parts, x.contiguous(), group=self.group)
        x_u, x_c = parts
        return x_u + self.scale * (x_c - x_u)
