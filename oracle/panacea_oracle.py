"""CPU oracle of the Panacea denoising hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 *functional* restatement of the reference's per-step eps_theta evaluation
(`OpenAIWrapperControlLDM3D -> ControlNet3D -> ControlledUNetModel3D`).  It is driven only by a flat
state dict (reference parameter names) and a few hyper-parameters, so it shares no code with the
product modules in `panacea_amd/`.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it — as the checker / reported baseline, never as a compute path.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle
is pinned against the reference ITSELF, imported in the build container by `oracle/gen_golden.py`;
the resulting vectors are committed under `tests/golden/` and `tests/test_oracle_golden.py` replays
them everywhere.  Third-party arithmetic outside the reference tree — xformers 0.0.16
`memory_efficient_attention` (attention.py:363,469,590) — is restated as softmax(q k^T d^-1/2) v.

Every function cites the reference lines it follows (paths relative to the reference root,
`sgm/modules/...`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
TRACE: Optional[dict] = None     # set to a dict to record every top-level block output (tests only)


@dataclass
class OracleConfig:
    num_frames: int = 8
    model_channels: int = 320
    num_head_channels: int = 64
    spatial_only_attn_type: Optional[str] = "intra-view"   # None -> plain self-attention (BASELINE config 1)
    insert_crossview: bool = True
    control_scales: float = 1.0
    faithful_temporal_context: bool = False   # True: project text K/V once per pixel like attention.py:1122-1125


# ----------------------------------------------------------------------------------------------
# helpers  (diffusionmodules/util.py)
# ----------------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """util.py:224-248 — [cos | sin], fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None].to(timesteps.device)
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def temporal_pos_embedding(pos_len: int, dim: int) -> torch.Tensor:
    """attention.py:1140-1159 — including its integer cast (quirk Q2): the frequency vector is cast to
    int64, which keeps only its first entry (=1), so column 0 = sin(p), column 1 = cos(p), every other
    even column = 0 and odd column = 1."""
    freq = 1.0 / torch.pow(torch.tensor(10000.0), torch.arange(dim // 2, dtype=torch.float32) / (dim / 2))
    freq = freq.to(torch.long)
    ang = (torch.arange(pos_len, dtype=torch.long)[:, None] * freq[None, :]).float()
    emb = torch.zeros(pos_len, dim, dtype=torch.float32)
    emb[:, 0::2] = torch.sin(ang)
    emb[:, 1::2] = torch.cos(ang)
    return emb


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _conv(sd: SD, p: str, x: torch.Tensor, stride: int = 1) -> torch.Tensor:
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=w.shape[-1] // 2)


# ----------------------------------------------------------------------------------------------
# attention  (attention.py)
# ----------------------------------------------------------------------------------------------
def _sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v on (b, n, heads*d) tensors — attention.py:264-284 / xformers FMHA."""
    b, n, c = q.shape
    d = c // heads
    qh = q.view(b, n, heads, d).transpose(1, 2)
    kh = k.view(b, k.shape[1], heads, d).transpose(1, 2)
    vh = v.view(b, v.shape[1], heads, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), vh)
    return o.transpose(1, 2).reshape(b, n, c)


def cross_attention(sd: SD, p: str, x: torch.Tensor, context: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """CrossAttention.forward, attention.py:229-291 (self-attention when context is None)."""
    ctx = x if context is None else context
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    return _lin(sd, p + ".to_out.0", _sdpa(q, k, v, heads))


def _view_attention(sd: SD, p: str, x: torch.Tensor, heads: int, inter: bool) -> torch.Tensor:
    """MemoryEfficientIntraViewAttention.forward (attention.py:407-489) and
    MemoryEfficientInterViewAttentionTwo.forward (attention.py:518-610)."""
    b, n, c = x.shape
    q_all, k_all, v_all = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", x), _lin(sd, p + ".to_v", x)
    H = int(math.sqrt(n // 12))                     # attention.py:428 / 537 — 6 views of aspect 1:2
    W = n // H
    if H * W != n or W % 6:
        raise ValueError(f"token count {n} is not a 6-view panorama (H={H})")
    width = W // 6
    grid = lambda t: t.view(b, H, W, c)
    qg, kg, vg = grid(q_all), grid(k_all), grid(v_all)
    outs = []
    for i in range(0, W, width):
        q = qg[:, :, i:i + width].reshape(b, H * width, c)
        if not inter:
            ks, vs = kg[:, :, i:i + width], vg[:, :, i:i + width]
        elif 0 < i < 6 * width:
            # attention.py:549-551.  For the last view (i = 5*width) the "right" slice
            # [i+width : i+2*width] is empty: view 5 attends to view 4 only (quirk Q1).
            ks = torch.cat([kg[:, :, i - width:i], kg[:, :, i + width:i + 2 * width]], dim=2)
            vs = torch.cat([vg[:, :, i - width:i], vg[:, :, i + width:i + 2 * width]], dim=2)
        else:                                        # i == 0: [view 5, view 1]  (attention.py:553-555)
            ks = torch.cat([kg[:, :, 5 * width:W], kg[:, :, width:2 * width]], dim=2)
            vs = torch.cat([vg[:, :, 5 * width:W], vg[:, :, width:2 * width]], dim=2)
        k = ks.reshape(b, -1, c)
        v = vs.reshape(b, -1, c)
        outs.append(_sdpa(q, k, v, heads).view(b, H, width, c))
    out = torch.cat(outs, dim=2).reshape(b, n, c)
    return _lin(sd, p + ".to_out.0", out)


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """FeedForward with GEGLU, attention.py:91-117 (gate = second half, erf GELU)."""
    h, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", h * F.gelu(gate))


def basic_transformer_block(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int,
                            attn1_kind: Optional[str]) -> torch.Tensor:
    """BasicTransformerBlock._forward, attention.py:726-747."""
    h = _ln(sd, p + ".norm1", x)
    if attn1_kind == "intra-view":
        a = _view_attention(sd, p + ".attn1", h, heads, inter=False)
    elif attn1_kind == "inter-view":
        a = _view_attention(sd, p + ".attn1", h, heads, inter=True)
    else:
        a = cross_attention(sd, p + ".attn1", h, None, heads)
    x = a + x
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_temporal_transformer(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """SpatialTemporalTransformer.forward, attention.py:1064-1134 (use_linear=True, depth 1..n)."""
    bt, c, h, w = x.shape
    T = cfg.num_frames
    heads = c // cfg.num_head_channels
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(bt, h * w, c)
    img = lambda t: t.view(bt, h, w, c).permute(0, 3, 1, 2)

    def depth(prefix: str) -> int:
        d = 0
        while f"{p}.{prefix}.{d}.norm1.weight" in sd:
            d += 1
        return d

    # intra-view (or plain) spatial branch — :1069-1085
    x_in = x
    t = _lin(sd, p + ".proj_in", tok(_gn(sd, p + ".norm", x, 1e-6)))
    for i in range(depth("transformer_blocks")):
        t = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", t, context, heads, cfg.spatial_only_attn_type)
    x = img(_lin(sd, p + ".proj_out", t)) + x_in

    # cross-view branch — :1087-1104
    if cfg.insert_crossview:
        x_in = x
        t = _lin(sd, p + ".proj_in_crossview", tok(_gn(sd, p + ".norm_crossview", x, 1e-6)))
        for i in range(depth("transformer_blocks_crossview")):
            t = basic_transformer_block(sd, f"{p}.transformer_blocks_crossview.{i}", t, context, heads, "inter-view")
        x = img(_lin(sd, p + ".proj_out_crossview", t)) + x_in

    # temporal branch — :1106-1134
    x_in = x
    b = bt // T
    t = _lin(sd, p + ".proj_in_temporal", tok(_gn(sd, p + ".norm_temporal", x, 1e-6)))
    t = t.view(b, T, h * w, c).permute(0, 2, 1, 3).reshape(b * h * w, T, c)          # (b h w) t c
    t = t + temporal_pos_embedding(T, c).to(t)                                        # :1117-1118
    ctx0 = context.view(b, T, context.shape[1], context.shape[2])[:, 0]               # :1122 frame 0 (Q3)
    for i in range(depth("transformer_blocks_temporal")):
        pb = f"{p}.transformer_blocks_temporal.{i}"
        if cfg.faithful_temporal_context:
            ctx = ctx0[:, None].expand(b, h * w, -1, -1).reshape(b * h * w, ctx0.shape[1], ctx0.shape[2])
            t = basic_transformer_block(sd, pb, t, ctx, heads, None)
        else:
            t = _temporal_block_dedup(sd, pb, t, ctx0, heads, h * w)
    t = t.view(b, h * w, T, c).permute(0, 2, 1, 3).reshape(bt, h * w, c)
    return x_in + img(_lin(sd, p + ".proj_out_temporal", t))


def _temporal_block_dedup(sd: SD, p: str, x: torch.Tensor, ctx0: torch.Tensor, heads: int, npix: int) -> torch.Tensor:
    """Same arithmetic as basic_transformer_block on the per-pixel repeated context, but the text
    K/V are projected once per sample instead of once per pixel (identical values, 1/npix the work)."""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads) + x
    h = _ln(sd, p + ".norm2", x)
    b = ctx0.shape[0]
    q = _lin(sd, p + ".attn2.to_q", h)
    k, v = _lin(sd, p + ".attn2.to_k", ctx0), _lin(sd, p + ".attn2.to_v", ctx0)       # (b, 77, c)
    T, c = x.shape[1], x.shape[2]
    o = _sdpa(q.view(b, npix * T, c), k, v, heads).view(b * npix, T, c)
    x = _lin(sd, p + ".attn2.to_out.0", o) + x
    return feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x


# ----------------------------------------------------------------------------------------------
# UNet blocks  (diffusionmodules/openaimodel.py)
# ----------------------------------------------------------------------------------------------
def _temporal_conv(sd: SD, p: str, h: torch.Tensor, T: int) -> torch.Tensor:
    """`h + conv1d(SiLU(GN(h)))` on "(b h w) c t" — openaimodel.py:505-515 / 533-539."""
    bt, c, H, W = h.shape
    b = bt // T
    z = h.view(b, T, c, H * W).permute(0, 3, 2, 1).reshape(b * H * W, c, T)
    z = F.conv1d(F.silu(_gn(sd, p + ".0", z, 1e-5)), sd[p + f".{_last(sd, p)}.weight"],
                 sd[p + f".{_last(sd, p)}.bias"], padding=1)
    z = z.view(b, H * W, c, T).permute(0, 3, 2, 1).reshape(bt, c, H, W)
    return h + z


def _last(sd: SD, p: str) -> int:
    """index of the conv inside an nn.Sequential prefix (2 without Dropout, 3 with)."""
    return 3 if (p + ".3.weight") in sd else 2


def resblock3d(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """ResBlock3D._forward, openaimodel.py:499-542 (no up/down, no scale-shift)."""
    T = cfg.num_frames
    h = _conv(sd, p + ".in_layers.2", F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)))
    h = _temporal_conv(sd, p + ".in_layers_temporal", h, T)
    h = h + _lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = _conv(sd, p + ".out_layers.3", F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)))
    h = _temporal_conv(sd, p + ".out_layers_temporal", h, T)
    skip = _conv(sd, p + ".skip_connection", x) if (p + ".skip_connection.weight") in sd else x
    return skip + h


def _run_block(sd: SD, p: str, h: torch.Tensor, emb: torch.Tensor, context: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    """TimestepEmbedSequential.forward (openaimodel.py:85-103): dispatch on what the child is, which the
    oracle reads off the parameter names."""
    j = 0
    while any(k.startswith(f"{p}.{j}.") for k in sd):
        q = f"{p}.{j}"
        if q + ".in_layers.0.weight" in sd:
            h = resblock3d(sd, q, h, emb, cfg)
        elif q + ".norm.weight" in sd:
            h = spatial_temporal_transformer(sd, q, h, context, cfg)
        elif q + ".op.weight" in sd:                              # Downsample, :161-201
            h = _conv(sd, q + ".op", h, stride=2)
        elif q + ".conv.weight" in sd:                            # Upsample, :129-142
            h = _conv(sd, q + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))
        elif q + ".weight" in sd:                                 # bare conv (stem / zero conv)
            h = _conv(sd, q, h)
        else:
            raise KeyError(f"unrecognised block {q}")
        j += 1
    if TRACE is not None:
        TRACE[p] = h.clone()
    return h


def _count(sd: SD, p: str) -> int:
    n = 0
    while any(k.startswith(f"{p}.{n}.") for k in sd):
        n += 1
    return n


def _time_embed(sd: SD, p: str, t: torch.Tensor, cfg: OracleConfig) -> torch.Tensor:
    e = timestep_embedding(t, cfg.model_channels)
    return _lin(sd, p + "time_embed.2", F.silu(_lin(sd, p + "time_embed.0", e)))


def _tile_context(context: torch.Tensor, T: int) -> torch.Tensor:
    """controlmodel.py:121-122 / 183-184."""
    return context[:, None].expand(-1, T, -1, -1).reshape(-1, context.shape[1], context.shape[2])


def controlnet_forward(sd: SD, cfg: OracleConfig, x: torch.Tensor, hint: torch.Tensor, t: torch.Tensor,
                       context: torch.Tensor, p: str = "controlnet.") -> List[torch.Tensor]:
    """ControlNet3D.forward, controlmodel.py:86-142."""
    emb = _time_embed(sd, p, t, cfg)
    g = hint
    for i, stride in zip(range(0, 16, 2), (1, 1, 2, 1, 2, 1, 2, 1)):         # controlmodel.py:43-59
        g = _conv(sd, f"{p}input_hint_block.{i}", g, stride=stride)
        if i != 14:
            g = F.silu(g)
    ctx = _tile_context(context, cfg.num_frames)
    outs, h = [], x
    for i in range(_count(sd, p + "input_blocks")):
        h = _run_block(sd, f"{p}input_blocks.{i}", h, emb, ctx, cfg)
        if i == 0:
            h = h + g                                                         # :125-129
        outs.append(_conv(sd, f"{p}zero_convs.{i}.0", h))
    h = _run_block(sd, p + "middle_block", h, emb, ctx, cfg)
    outs.append(_conv(sd, p + "middle_block_out.0", h))
    return [o * cfg.control_scales for o in outs]                             # :137-140


def unet_forward(sd: SD, cfg: OracleConfig, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor,
                 control: Optional[List[torch.Tensor]], p: str = "") -> torch.Tensor:
    """ControlledUNetModel3D.forward, controlmodel.py:160-202 (control=None: UNetModel3D.forward)."""
    emb = _time_embed(sd, p, t, cfg)
    ctx = _tile_context(context, cfg.num_frames)
    control = list(control) if control is not None else None
    hs, h = [], x
    for i in range(_count(sd, p + "input_blocks")):
        h = _run_block(sd, f"{p}input_blocks.{i}", h, emb, ctx, cfg)
        hs.append(h)
    h = _run_block(sd, p + "middle_block", h, emb, ctx, cfg)
    if control is not None:
        h = h + control.pop()
    for i in range(_count(sd, p + "output_blocks")):
        skip = hs.pop()
        if control is not None:
            skip = skip + control.pop()
        h = _run_block(sd, f"{p}output_blocks.{i}", torch.cat([h, skip], dim=1), emb, ctx, cfg)
    return _conv(sd, p + "out.2", F.silu(_gn(sd, p + "out.0", h, 1e-5)))


def wrapper_forward(sd: SD, cfg: OracleConfig, x: torch.Tensor, t: torch.Tensor, c: dict) -> torch.Tensor:
    """OpenAIWrapperControlLDM3D.forward, wrappers.py:37-70.  `sd` uses the names of the wrapped
    ControlledUNetModel3D (no `diffusion_model.` prefix)."""
    with torch.no_grad():
        xin = torch.cat([x, c["concat"]], dim=1).float() if "concat" in c else x.float()
        context = c["crossattn"].float()
        control = None
        if any(k.startswith("controlnet.") for k in sd):
            control = controlnet_forward(sd, cfg, xin, c["cond_feat"].float(), t, context)
        return unet_forward(sd, cfg, xin, t, context, control)
