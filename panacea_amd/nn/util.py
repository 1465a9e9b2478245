"""Host-side mirror of `sgm/modules/diffusionmodules/util.py` (the helpers the hot path imports)
plus the small glue between NCHW tensors at the API boundary and the resident token layout."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import engine as E
from ..engine import Act, Runtime


def checkpoint(func, inputs, params, flag):
    """util.py:153-221.  Activation checkpointing only matters for training; under no_grad the reference's
    CheckpointFunction just calls `func(*inputs)`, which is all that is kept."""
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """util.py:224-248 — sinusoidal embedding [cos | sin] in fp32, computed on the device of `timesteps`."""
    if repeat_only:
        return timesteps[:, None].expand(-1, dim)
    if max_period != 10000:
        raise NotImplementedError("max_period other than 10000 is not used on the Panacea path")
    F = timesteps.shape[0]
    be = E.backend()
    out = torch.zeros((F, dim), device=timesteps.device, dtype=torch.float32)
    be.timestep_embedding(timesteps.to(torch.int64).contiguous(), F, dim, E.timestep_freqs(dim, timesteps.device), out)
    return out


def zero_module(module):
    """util.py:251-257"""
    for p in module.parameters():
        p.detach().zero_()
    return module


def normalization(channels):
    """util.py:276-283 — parameter container; the arithmetic runs in pnc_groupnorm_*"""
    return nn.GroupNorm(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims == 1:
        return nn.Conv1d(*args, **kwargs)
    if dims == 2:
        return nn.Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims}")


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


# ---- API-boundary glue ------------------------------------------------------------------------
def runtime_for(x: torch.Tensor, num_frames: int, shard=None, vshard=None) -> Runtime:
    """`shard`: engine.FrameShard when x carries only this rank's num_frames / G frames of every sample; `vshard`:
    engine.ViewShard when x carries only this rank's band of views (W / G columns)"""
    F = x.shape[0]
    t_local = num_frames // (shard.G if shard is not None else 1)
    if t_local < 1 or F % t_local:
        raise ValueError(f"batch of {F} frames is not a multiple of the {t_local} frames per sample held by this rank")
    return Runtime(x.device, F // t_local, num_frames, shard, vshard)


def act_from_nchw(rt: Runtime, x: torch.Tensor) -> Act:
    """NCHW -> resident tokens (boundary plumbing for the per-module reference-compatible forwards; the
    network-level path converts its inputs with pnc_nchw_to_tokens_f16 instead)."""
    F, C, H, W = x.shape
    t = x.detach().permute(0, 2, 3, 1).reshape(F * H * W, C).to(torch.float32).contiguous()
    return Act(F, H, W, C, f32=t)
