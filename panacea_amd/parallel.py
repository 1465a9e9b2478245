"""Multi-GPU execution of the denoising path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X, "gloo" in the CPU tests).

What shards, and how.  Rank grid of one node: samples x CFG halves x frame groups,
rank = ((sample * cfg + half) * G + frame_group) * V + view_group  (`RankLayout`; frame groups and view groups compose).

* **samples** — the reference's own strategy (`inference.py:248-280`: DistributedSampler + a DDP wrapper used only
  for `.module`): independent units, no collective on the data path.  `replica_seed()` mirrors `inference.py:250`.
* **CFG halves** (cfg = 2) — inside `OpenAIWrapperControlLDM3D.forward` the unconditional and conditional halves of the
  guidance batch never interact (every op is per sample; GroupNorm statistics are per frame).  Only
  `VanillaCFG.__call__` (guiders.py:25-29) combines them.  `ShardedCFG` is a drop-in guider that gives each rank
  of a pair ONE half (half the frames per network evaluation) and exchanges the denoised halves with one all-gather of
  (T_local, 4, h, w) per step — 0.8 MB at the nuScenes shape; each peer pair has its own xGMI link.
* **frame groups** (G = 2 | 4) — rank g of a frame group holds frames [g*T/G, (g+1)*T/G) of its half: whole panoramic
  frames, so the 3x3 convs, the spatial GroupNorms and the intra-/cross-view attention (which couple the six VIEWS of a
  frame: attention.py:545-559, util.py:276-283) need nothing from another rank.  The cross-frame couplings — the two
  temporal GroupNorm + conv1d sites of every ResBlock3D (openaimodel.py:505-515, 533-539) and the temporal transformer
  branch of every STT (attention.py:1106-1134) — are pointwise per pixel across frames and run in the transposed
  sharding (all T frames of N/G pixels per rank): `engine.FrameShard.to_pixels / to_frames`, one all-to-all each.
  Bytes per rank and step, and the overlap plan, are in DESIGN.md §9.
* **view groups** (V = 2 | 3 | 6) — the alternative split of one half: rank v of a view group holds views
  [v*6/V, (v+1)*6/V) of every frame, a band of W/V columns of every map (`engine.ViewShard`).  The temporal sites are then
  local; what crosses ranks are the one-column halos of the 3x3 convs, the spatial GroupNorm statistics (96 floats per
  frame) and the keys / values of the two neighbouring views in the cross-view attention — neighbour-to-neighbour
  messages, each pair on its own xGMI link.  More, smaller messages than the frame exchange (DESIGN.md section 9 has the
  counts); it is the layout for V ranks when T / G frames per rank would drop below the temporal kernels' tile.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from .engine import FrameShard, ViewShard
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


def replica_seed(sample: int, base: int = 3407) -> int:
    """inference.py:250 — every replica (SAMPLE) draws its own noise.  With intra-sample sharding pass
    `RankLayout.sample`, never the rank: all ranks of one sample must draw the same latent."""
    return base + sample


def cfg_pair_groups(world: int) -> List[Optional[dist.ProcessGroup]]:
    """One 2-rank group per sample: ranks (2k, 2k+1).  Every rank must call this (collective group creation)."""
    if world % 2:
        raise ValueError("CFG sharding needs an even number of ranks")
    return [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]


@dataclass(frozen=True)
class RankLayout:
    """rank = ((sample * cfg + half) * frames + frame_group) * views + view_group"""
    world: int
    rank: int
    cfg: int = 1            # 1: both CFG halves on every rank; 2: one half per rank
    frames: int = 1         # frame groups per half (G)
    views: int = 1          # view groups per half (V): 1, 2, 3 or 6; composes with frame groups (SURVEY 8e's 8-GPU grid:
                            # cfg 2 x 2 view groups x 2 frame groups — a rank then holds T/G frames of a band of W/V columns)

    def __post_init__(self):
        if self.cfg not in (1, 2) or self.frames < 1 or self.views not in (1, 2, 3, 6) \
                or self.world % (self.cfg * self.frames * self.views):
            raise ValueError(f"{self.world} ranks do not factor into cfg {self.cfg} x frame groups {self.frames} x "
                             f"view groups {self.views}")

    @property
    def per_sample(self) -> int:
        return self.cfg * self.frames * self.views

    @property
    def samples(self) -> int:
        return self.world // self.per_sample

    @property
    def sample(self) -> int:
        return self.rank // self.per_sample

    @property
    def half(self) -> int:
        return (self.rank // (self.frames * self.views)) % self.cfg

    @property
    def frame_group(self) -> int:
        return (self.rank // self.views) % self.frames

    @property
    def view_group(self) -> int:
        return self.rank % self.views

    def frame_group_ranks(self, sample: int, half: int, view_group: int = 0) -> List[int]:
        """the ranks that hold the frame groups of one (sample, half, view band)"""
        base = (sample * self.cfg + half) * self.frames * self.views
        return [base + g * self.views + view_group for g in range(self.frames)]

    def view_group_ranks(self, sample: int, half: int, frame_group: int = 0) -> List[int]:
        """the ranks that hold the view bands of one (sample, half, frame group)"""
        base = (sample * self.cfg + half) * self.frames * self.views
        return [base + frame_group * self.views + v for v in range(self.views)]

    def cfg_pair_ranks(self, sample: int, part: int) -> List[int]:
        """the ranks holding the same frame group / view group (`part`) of the two halves of a sample"""
        return [(sample * self.cfg + h) * self.frames * self.views + part for h in range(self.cfg)]

    @property
    def name(self) -> str:
        parts = [f"replica x{self.samples}"] if self.samples > 1 else []
        if self.cfg > 1:
            parts.append("cfg x2")
        if self.frames > 1:
            parts.append(f"frames x{self.frames}")
        if self.views > 1:
            parts.append(f"views x{self.views}")
        return " . ".join(parts) or "single"


def layout_for(world: int, rank: int, parallelism: str, num_frames: int = 8) -> RankLayout:
    """`replica` (one sample per rank) | `cfg` (one sample per rank pair) | `cfg+frames` (one sample over up to 8 ranks:
    2 CFG halves x min(4, world/2) frame groups; with a single rank it degenerates to `replica`)."""
    if parallelism == "replica" or world == 1:
        return RankLayout(world, rank)
    if parallelism == "cfg":
        return RankLayout(world, rank, cfg=2)
    def frame_groups(per):                      # largest G <= 4 dividing the ranks available to a half AND the clip's frames
        return [g for g in (4, 2, 1) if per % g == 0 and num_frames % g == 0][0]
    if parallelism == "cfg+frames":
        if world % 2:
            raise ValueError(f"cfg+frames needs an even number of ranks, got {world}")
        return RankLayout(world, rank, cfg=2, frames=frame_groups(world // 2))
    if parallelism == "frames":
        return RankLayout(world, rank, cfg=1, frames=frame_groups(world))
    if parallelism == "cfg+views+frames":
        # SURVEY 8(e)'s 8-GPU grid: 2 CFG halves x 2 view groups (3 views each) x world/4 frame groups
        if world % 4 or world < 8:
            raise ValueError(f"cfg+views+frames needs a multiple of 4 ranks >= 8 (2 halves x 2 view groups x G frame groups), got {world}")
        # G: the largest frame-group count <= 4 that divides BOTH the ranks of a (half, view group) and the clip's frames, so that
        # 2 x 2 x G factors the sample's ranks and every rank holds whole frames (world 12 -> G = 1 x 3 samples, not G = 3)
        return RankLayout(world, rank, cfg=2, frames=frame_groups(world // 4), views=2)
    if parallelism in ("views", "cfg+views"):
        cfg = 2 if parallelism == "cfg+views" else 1
        fit = [v for v in (6, 3, 2) if (world // cfg) % v == 0 and world >= cfg * v]
        if not fit:
            raise ValueError(f"{world} ranks do not hold a view split (2, 3 or 6 view groups x {cfg} CFG halves)")
        return RankLayout(world, rank, cfg=cfg, views=fit[0])
    raise ValueError(f"unknown parallelism {parallelism!r}")


class Groups:
    """The process groups of this rank.  Group creation is collective: EVERY rank constructs this with the same layout
    (all groups are created by all ranks, in the same order)."""

    def __init__(self, layout: RankLayout, side: bool = False):
        """`side` (opt-in since round 6: ADVICE r5 — two communicators driven from two streams are only safe while every rank issues
        in the same host order and both collectives' kernels can co-reside, and no multi-GPU RCCL node has validated that yet; the
        default keeps both networks on ONE stream over ONE set of communicators in the sharded layouts, as rounds 2-4 did, and
        creates half the communicators): create every frame / view group TWICE.  The second set belongs to the ControlNet (`frame_shard(side=True)`,
        `view_shard(side=True)`, `apply_*_shard(net, shard, side_shard)`): with communicators of its own the ControlNet's collectives
        keep ONE order per communicator whatever the interleaving with the UNet's, so the two networks may run on two HIP streams in
        the sharded layouts as they do on one GPU (round 5; rounds 2-4 put both on one stream there)."""
        self.layout = layout
        self.frame_group = self.view_group = self.cfg_pair = None
        self.frame_group_side = self.view_group_side = None
        self._view_shard: Optional[ViewShard] = None
        self._view_shard_side: Optional[ViewShard] = None
        for smp in range(layout.samples):
            for h in range(layout.cfg):
                for vg in range(layout.views if layout.frames > 1 else 0):
                    ranks = layout.frame_group_ranks(smp, h, vg)
                    g = dist.new_group(ranks)
                    g2 = dist.new_group(ranks) if side else None
                    if layout.rank in ranks:
                        self.frame_group, self.frame_group_side = g, g2
                for fg in range(layout.frames if layout.views > 1 else 0):
                    ranks = layout.view_group_ranks(smp, h, fg)
                    g = dist.new_group(ranks)
                    g2 = dist.new_group(ranks) if side else None
                    if layout.rank in ranks:
                        self.view_group, self.view_group_side = g, g2
            for fg in range(layout.frames * layout.views):
                ranks = layout.cfg_pair_ranks(smp, fg)
                if layout.cfg > 1:
                    g = dist.new_group(ranks)
                    if layout.rank in ranks:
                        self.cfg_pair = g

    def frame_shard(self, side: bool = False) -> Optional[FrameShard]:
        lo = self.layout
        if lo.frames <= 1 or (side and self.frame_group_side is None):
            return None
        return FrameShard(lo.frames, lo.frame_group, self.frame_group_side if side else self.frame_group)

    def view_shard(self, side: bool = False) -> Optional[ViewShard]:
        lo = self.layout
        if lo.views <= 1:
            return None
        if side:
            if self.view_group_side is None:
                return None
            if self._view_shard_side is None:
                self._view_shard_side = ViewShard(lo.views, lo.view_group, self.view_group_side)
            return self._view_shard_side
        if self._view_shard is None:
            self._view_shard = ViewShard(lo.views, lo.view_group, self.view_group)
        return self._view_shard

    def guider(self, scale: float) -> VanillaCFG:
        return ShardedCFG(scale, self.cfg_pair, self.layout.half) if self.layout.cfg > 1 else VanillaCFG(scale)


def apply_frame_shard(network, shard: Optional[FrameShard], side_shard: Optional[FrameShard] = None):
    """Tell the network (OpenAIWrapperControlLDM3D or ControlledUNetModel3D) that its batches carry this rank's frame
    group only.  `side_shard`: the same shard over the ControlNet's own process group (`Groups.frame_shard(side=True)`): with it
    the ControlNet runs on its side stream in the sharded layout too."""
    model = getattr(network, "diffusion_model", network)
    model.frame_shard = shard
    if hasattr(model, "controlnet"):
        model.controlnet.frame_shard = side_shard if (shard is not None and side_shard is not None) else shard
    return network


def apply_view_shard(network, shard: Optional[ViewShard], side_shard: Optional[ViewShard] = None):
    """Tell the network that its batches carry this rank's band of views only (W / V columns of the latent, of `concat`
    and of the BEV hint).  `side_shard`: see apply_frame_shard."""
    model = getattr(network, "diffusion_model", network)
    model.view_shard = shard
    if hasattr(model, "controlnet"):
        model.controlnet.view_shard = side_shard if (shard is not None and side_shard is not None) else shard
    return network


def local_views(t: torch.Tensor, layout: RankLayout) -> torch.Tensor:
    """this rank's band of views of an NCHW map (latent, `concat`, or the 8x larger hint): columns [v W/V, (v+1) W/V)"""
    if layout.views == 1:
        return t
    W = t.shape[-1]
    if W % layout.views:
        raise ValueError(f"map width {W} does not split into {layout.views} view groups")
    wl = W // layout.views
    return t[..., layout.view_group * wl:(layout.view_group + 1) * wl].contiguous()


def gather_views(x_local: torch.Tensor, groups: "Groups") -> torch.Tensor:
    """the whole panorama from the view group (end of the schedule: hand the latent to the first-stage decoder)"""
    return x_local if groups.layout.views == 1 else groups.view_shard().gather_width(x_local)


def local_frames(t: torch.Tensor, layout: RankLayout, num_frames: int) -> torch.Tensor:
    """this rank's frames of a per-frame tensor whose leading dimension is (samples_in_t * num_frames)"""
    if layout.frames == 1:
        return t
    if num_frames % layout.frames:
        raise ValueError(f"{num_frames} frames do not split over {layout.frames} frame groups")
    tl = num_frames // layout.frames
    v = t.view(t.shape[0] // num_frames, num_frames, *t.shape[1:])
    return v[:, layout.frame_group * tl:(layout.frame_group + 1) * tl].reshape(-1, *t.shape[1:]).contiguous()


def shard_conditioning(cond: Dict, layout: RankLayout, num_frames: int) -> Dict:
    """per-frame conditioning (`concat`, `cond_feat`) cut to this rank's frame group / view band; per-sample entries
    untouched"""
    out = dict(cond)
    for k in ("concat", "cond_feat"):
        if k in out:
            out[k] = local_views(local_frames(out[k], layout, num_frames), layout)
    return out


def gather_frames(x_local: torch.Tensor, groups: "Groups", num_frames: int) -> torch.Tensor:
    """all frames of the sample from the frame group (end of the schedule: hand the latent to the first-stage decoder)"""
    lo = groups.layout
    if lo.frames == 1:
        return x_local
    tl = num_frames // lo.frames
    if x_local.shape[0] % tl:
        raise ValueError(f"{x_local.shape[0]} local frames are not a multiple of the {tl} frames per sample of this rank")
    nb = x_local.shape[0] // tl                 # samples in the batch
    parts = [torch.empty_like(x_local) for _ in range(lo.frames)]
    dist.all_gather(parts, x_local.contiguous(), group=groups.frame_group)
    # parts[g] holds rows (b, t_local) of frame group g; the sample's frames in global order are (b, g, t_local)
    # — the order local_frames() cut them in and FrameShard.gather_rows() restores (a plain cat interleaves
    # the samples for more than one sample per rank)
    return torch.stack([q.view(nb, tl, *q.shape[1:]) for q in parts], dim=1).reshape(nb * lo.frames * tl, *x_local.shape[1:])


class ShardedCFG(VanillaCFG):
    """VanillaCFG over a pair of ranks: one evaluates the unconditional half, the other the conditional half (the
    reference's batch order, uncond first: guiders.py:36,40).  BOTH ranks of a pair must hold the SAME latent x and sigma:
    seed the pair with `replica_seed(layout.sample)` (not the rank) or broadcast x from the first rank of the pair —
    `check_pair_consistency()` asserts it."""

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

    def check_pair_consistency(self, x: torch.Tensor, sigma: torch.Tensor):
        """Debug guard for the first step: both ranks of the pair must start from bit-identical (x, sigma) — a caller
        that seeds per RANK would silently combine eps of two different latents."""
        digest = torch.stack([x.double().sum(), x.double().abs().sum(), sigma.double().sum()]).to(x.device)
        parts = [torch.empty_like(digest), torch.empty_like(digest)]
        dist.all_gather(parts, digest, group=self.group)
        if not torch.equal(parts[0], parts[1]):
            raise RuntimeError("ShardedCFG: the two ranks of a CFG pair hold different latents / sigmas; seed the pair with "
                               "replica_seed(layout.sample) or broadcast the initial latent over the pair group")
