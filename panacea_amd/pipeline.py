"""Latents-in, frames-out composition of the pieces this package owns (hot path + f1 + f2 + f4):

    cond / uc  ->  EulerEDMSampler + VanillaCFG + DiscreteDenoiser around the ControlNet-UNet (step invariants hoisted)
               ->  z / scale_factor  ->  FirstStageDecoder  ->  frames in [-1, 1]  (-> checkpoint.save_view_frames / save_gif)

which is what `DiffusionEngine3D.sample` + `decode_first_stage` do around the network (diffusion.py:138-151, 242-249;
`scale_factor` 0.18215, inference_nuscenes.yaml:5).  The text / image conditioners are not part of it:
`cond` and `uc` arrive as tensors.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import sampling

SCALE_FACTOR = 0.18215


def sample_frames(network, first_stage, cond: Dict[str, torch.Tensor], uc: Dict[str, torch.Tensor],
                  noise: torch.Tensor, num_steps: int = 25, cfg_scale: float = 5.0, hoist: bool = True,
                  scale_factor: float = SCALE_FACTOR) -> torch.Tensor:
    """noise: (T, 4, h, w) unit-variance latents of ONE sample; returns (T, 3, 8h, 8w) frames."""
    dev = noise.device
    den = sampling.DiscreteDenoiser().to(dev)
    smp = sampling.EulerEDMSampler(num_steps, guider=sampling.VanillaCFG(cfg_scale), device=dev)
    with torch.no_grad():
        z = smp(sampling.BoundDenoiser(den, network), noise, cond, uc, network=network if hoist else None)
        return first_stage.decode(z / scale_factor)
