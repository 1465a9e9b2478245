"""Named network configurations (constructor kwargs of ControlledUNetModel3D / ControlNet3D).

`FULL` is the reference's Panacea+ stage-2 network (configs/inference_nuscenes.yaml:30-71);
`TINY` keeps every code path of it (two levels, intra-view + cross-view + temporal attention,
ControlNet, Down/Upsample, Cin != Cout skips) at a size the CPU oracle finishes in a second;
`PLAIN1` is BASELINE config 1 — the only way the reference accepts a single-view square latent
(spatial_only_attn_type=None, insert_crossview=False, num_frames=1).
"""
from __future__ import annotations

import copy

FULL = dict(
    insert_crossview=True, spatial_only_attn_type="intra-view", use_checkpoint=True, use_fp16=True,
    in_channels=8, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
    channel_mult=[1, 2, 4, 4], num_head_channels=64, use_spatial_transformer=True,
    use_linear_in_transformer=True, transformer_depth=1, context_dim=1024, legacy=False,
    num_frames=8, alpha=1,
)

TINY = dict(
    insert_crossview=True, spatial_only_attn_type="intra-view", use_checkpoint=True,
    in_channels=8, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
    channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True,
    use_linear_in_transformer=True, transformer_depth=1, context_dim=64, legacy=False,
    num_frames=2, alpha=1,
)

PLAIN1 = dict(
    insert_crossview=False, spatial_only_attn_type=None, use_checkpoint=True,
    in_channels=8, model_channels=64, attention_resolutions=[2, 1], num_res_blocks=1,
    channel_mult=[1, 2], num_head_channels=64, use_spatial_transformer=True,
    use_linear_in_transformer=True, transformer_depth=1, context_dim=64, legacy=False,
    num_frames=1, alpha=1,
)

CONFIGS = {"full": FULL, "tiny": TINY, "plain1": PLAIN1}

# (B, T, h, w) of the synthetic step inputs used with each configuration
SHAPES = {
    "full": (2, 8, 32, 384),      # BASELINE config 3: CFG batch 2 x 8 frames, 6 views x (32 x 64) latent
    "tiny": (2, 2, 8, 96),
    "plain1": (1, 1, 16, 16),
}


def get(name: str) -> dict:
    return copy.deepcopy(CONFIGS[name])


def with_frames(cfg: dict, num_frames: int) -> dict:
    c = copy.deepcopy(cfg)
    c["num_frames"] = num_frames
    return c
