"""CPU oracle of the first-stage decoder — TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32, functional over a flat state dict with the reference's parameter names
(`decoder.*`, `post_quant_conv.*`).  Each function cites the reference lines it restates
(`sgm/modules/diffusionmodules/model.py`, `sgm/models/autoencoder.py`).  Pinned by `oracle/gen_golden_vae.py`,
which runs the reference's own `Decoder` in the build container on deterministic synthetic weights and commits the
outputs as `tests/golden/vae_tiny.npz`.  Only tests/ may import this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class VaeConfig:
    ch: int = 128
    out_ch: int = 3
    ch_mult: List[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_res_blocks: int = 2
    z_channels: int = 4
    embed_dim: int = 4
    attn_mid: bool = True                     # attn_type "vanilla": AttnBlock in the middle (model.py:943)


def _gn(x, sd, pre):
    """Normalize = GroupNorm(32, eps 1e-6, affine)  (model.py:59-62)"""
    return F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], 1e-6)


def _swish(x):
    """nonlinearity (model.py:54-57)"""
    return x * torch.sigmoid(x)


def resnet_block(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward with temb = None (model.py:176-196)"""
    h = F.conv2d(_swish(_gn(x, sd, pre + ".norm1")), sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(h, sd, pre + ".norm2")), sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"], padding=1)
    if pre + ".nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[pre + ".nin_shortcut.weight"], sd[pre + ".nin_shortcut.bias"])
    elif pre + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[pre + ".conv_shortcut.weight"], sd[pre + ".conv_shortcut.bias"], padding=1)
    return x + h


def attn_block(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward: single head over the h*w tokens, scale C^-1/2 (model.py:393-415)"""
    b, c, hh, ww = x.shape
    h = _gn(x, sd, pre + ".norm")
    q, k, v = (F.conv2d(h, sd[f"{pre}.{n}.weight"], sd[f"{pre}.{n}.bias"]).flatten(2).transpose(1, 2) for n in "qkv")
    p = torch.softmax(q @ k.transpose(1, 2) * (float(c) ** -0.5), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, c, hh, ww)
    return x + F.conv2d(o, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])


def decoder_forward(sd: Dict[str, torch.Tensor], cfg: VaeConfig, z: torch.Tensor, pre: str = "decoder", trace=None):
    """Decoder.forward (model.py:993-1026)"""
    h = F.conv2d(z, sd[pre + ".conv_in.weight"], sd[pre + ".conv_in.bias"], padding=1)
    h = resnet_block(sd, pre + ".mid.block_1", h)
    if cfg.attn_mid:
        h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    if trace is not None:
        trace["mid"] = h
    nres = len(cfg.ch_mult)
    for i_level in reversed(range(nres)):
        for i_block in range(cfg.num_res_blocks + 1):
            h = resnet_block(sd, f"{pre}.up.{i_level}.block.{i_block}", h)
        if i_level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")                  # Upsample (model.py:74-77)
            h = F.conv2d(h, sd[f"{pre}.up.{i_level}.upsample.conv.weight"], sd[f"{pre}.up.{i_level}.upsample.conv.bias"],
                         padding=1)
        if trace is not None:
            trace[f"up.{i_level}"] = h
    h = _swish(_gn(h, sd, pre + ".norm_out"))
    return F.conv2d(h, sd[pre + ".conv_out.weight"], sd[pre + ".conv_out.bias"], padding=1)


def decode(sd: Dict[str, torch.Tensor], cfg: VaeConfig, z: torch.Tensor, trace=None) -> torch.Tensor:
    """AutoencoderKL.decode: decoder(post_quant_conv(z))  (autoencoder.py:364-367)"""
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder_forward(sd, cfg, z, trace=trace)


def encoder_forward(sd: Dict[str, torch.Tensor], cfg: VaeConfig, x: torch.Tensor, pre: str = "encoder", trace=None):
    """Encoder.forward (model.py:852-880); Downsample = F.pad (0,1,0,1) + conv stride 2 padding 0 (model.py:108-112)"""
    h = F.conv2d(x, sd[pre + ".conv_in.weight"], sd[pre + ".conv_in.bias"], padding=1)
    nres = len(cfg.ch_mult)
    for i_level in range(nres):
        for i_block in range(cfg.num_res_blocks):
            h = resnet_block(sd, f"{pre}.down.{i_level}.block.{i_block}", h)
        if i_level != nres - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"{pre}.down.{i_level}.downsample.conv.weight"],
                         sd[f"{pre}.down.{i_level}.downsample.conv.bias"], stride=2)
        if trace is not None:
            trace[f"down.{i_level}"] = h
    h = resnet_block(sd, pre + ".mid.block_1", h)
    if cfg.attn_mid:
        h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    h = _swish(_gn(h, sd, pre + ".norm_out"))
    return F.conv2d(h, sd[pre + ".conv_out.weight"], sd[pre + ".conv_out.bias"], padding=1)


def encode_moments(sd: Dict[str, torch.Tensor], cfg: VaeConfig, x: torch.Tensor, trace=None) -> torch.Tensor:
    """AutoencoderKL.encode up to the posterior parameters: quant_conv(encoder(x))  (autoencoder.py:355-362)"""
    h = encoder_forward(sd, cfg, x, trace=trace)
    return F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
