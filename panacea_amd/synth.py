"""Deterministic, platform-independent synthetic weights and inputs.

There is no network (the published 9 GB checkpoint is unavailable), so parity tests, the smoke
test and `bench.py` all run on weights generated here: a PCG64 stream seeded from CRC32(name), scaled
so activations stay O(1), and rounded to fp16-representable values (the HIP path stores weights in
fp16; the oracle then sees exactly the same numbers and weight quantisation is not counted as
error).  Tensors the reference zero-initialises (`proj_out*`, `out_layers.3`, temporal convs,
zero-convs, `out.2`, hint-stem tail; openaimodel.py:418,454,468,1251, attention.py:1040-1059,
controlmodel.py:58,83) get non-zero values too — a freshly built reference network outputs
exactly 0, which would make parity vacuous.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_RESIDUAL_OUT = ("proj_out", "out_layers.3", "in_layers_temporal.2", "out_layers_temporal.3",
                 "zero_convs", "middle_block_out", "input_hint_block.14", "to_out.0", "ff.net.2")


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) + 0x9E3779B1 * salt) & 0xFFFFFFFF))


def _fp16_round(a: np.ndarray) -> np.ndarray:
    return a.astype(np.float16).astype(np.float32)


TAIL_PERIOD, TAIL_PHASE = 64, 5       # heavy-tail weight sets: output channels c with c % 64 == 5 of every residual-out tensor


def synth_tensor(name: str, shape: Tuple[int, ...], salt: int = 0, tail: float = 0.0) -> torch.Tensor:
    """One parameter, from its NAME and SHAPE only.

    `tail` > 0 (a power of two, so the values stay fp16-representable) multiplies the output rows c % 64 == 5 of every tensor that
    writes into the residual stream: a few "massive activation" channels, as trained diffusion / transformer checkpoints have
    them, carry the stream to |v| = 10^2 .. 10^3 — the regime where the e4m3 lo plane of an fp16-rounded operand clamps
    (include/panacea_hip.h: |v| >= 512) and a GroupNorm group is dominated by one channel.  Default 0: the weight sets of every
    earlier pin are unchanged."""
    g = _rng(name, salt)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = (len(shape) == 1 and leaf in ("weight", "bias")
               and any(t in name for t in (".norm", "in_layers.0", "out_layers.0", "in_layers_temporal.0",
                                           "out_layers_temporal.0", "out.0")))
    if is_norm:
        a = g.standard_normal(shape, dtype=np.float32) * 0.1 + (1.0 if leaf == "weight" else 0.0)
    elif leaf == "bias":
        a = g.standard_normal(shape, dtype=np.float32) * 0.02
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        gain = 0.5 if any(t in name for t in _RESIDUAL_OUT) else 1.0
        a = g.standard_normal(shape, dtype=np.float32) * (gain / np.sqrt(max(fan_in, 1)))
        if tail > 0.0 and gain != 1.0 and len(shape) > 1:
            a[TAIL_PHASE::TAIL_PERIOD] *= np.float32(tail)
    return torch.from_numpy(_fp16_round(a))


def synth_state_dict(manifest: Dict[str, Iterable[int]], salt: int = 0, threads: int = 8, tail: float = 0.0) -> Dict[str, torch.Tensor]:
    """Every tensor is generated from its own (name, shape) stream, so the result does not depend on the
    thread count; numpy's Generator releases the GIL while sampling."""
    items = list(manifest.items())
    if threads <= 1 or len(items) < 64:
        return {k: synth_tensor(k, tuple(v), salt, tail) for k, v in items}
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(threads) as ex:
        vals = list(ex.map(lambda kv: synth_tensor(kv[0], tuple(kv[1]), salt, tail), items))
    return {k: v for (k, _), v in zip(items, vals)}


def synth_inputs(B: int, T: int, h: int, w: int, context_dim: int = 1024, hint_channels: int = 19,
                 t_index: int = 999, salt: int = 0, hint_scale: int = 8) -> dict:
    """Synthetic step inputs of the nuScenes shape: CFG batch B (uncond half first),
    T frames, latent h x w (6 views along the width), BEV-layout hint at `hint_scale` x the latent."""
    F = B * T

    def n(name, shape, scale=1.0):
        return torch.from_numpy(_rng("input." + name, salt).standard_normal(shape, dtype=np.float32) * scale)
    x = n("x", (F, 4, h, w))
    concat = n("concat", (F, 4, h, w), 0.18215 * 5)
    crossattn = n("crossattn", (B, 77, context_dim))
    hint = torch.from_numpy(_rng("input.cond_feat", salt).random((T, hint_channels, hint_scale * h, hint_scale * w),
                                                                 dtype=np.float32))
    cond_feat = hint.repeat(B, 1, 1, 1)            # the uc / c halves share the layout hint
    t = torch.full((F,), int(t_index), dtype=torch.int64)
    return {"x": x, "t": t, "concat": concat, "crossattn": crossattn, "cond_feat": cond_feat}


def yaml_exact_step0_inputs(T: int, h: int, w: int, context_dim: int = 1024, salt: int = 0,
                            share_noise_level: float = 0.07) -> dict:
    """The network-level inputs of sampler step 0 of BASELINE config 5 (configs/inference_nuscenes.yaml): `concat` is the
    `final_cond_zero` pattern (nuscenes_datasets_video.py:559-572: the conditioning image sits in the LAST frame, every other
    frame encodes a zero image, i.e. one constant latent per channel) for BOTH CFG halves, the latent of every frame is
    `randn + share_noise_level * concat[-1]` (diffusion.py:242-249) — at step 0 the denoiser's c_in exactly undoes the
    sampler's sqrt(1 + sigma_0^2) — and t = 999.  Same tensor convention as `synth_inputs` (uncond half first)."""
    inp = synth_inputs(2, T, h, w, context_dim=context_dim, t_index=999, salt=salt)
    cc = inp["concat"][T:].clone()
    cc[:-1] = cc[0:1].mean(dim=(2, 3), keepdim=True).expand(-1, -1, h, w)
    x0 = inp["x"][T:] + cc[-1].unsqueeze(0) * share_noise_level
    inp["concat"] = torch.cat([cc, cc], dim=0)
    inp["x"] = torch.cat([x0, x0], dim=0)
    return inp
