"""Decomposed-4D attention on MI355X — host-side mirror of `sgm/modules/attention.py`.

Same class names, constructor arguments and parameter names as the reference (so its checkpoints
and YAML `target:` strings work unchanged); the forwards are new: they enqueue the hand-written
gfx950 kernels of libpanacea_hip.so on channels-last token matrices (`panacea_amd.engine`).

  reference class (attention.py)                 -> what runs here
  SpatialTemporalTransformer :898-1134           -> three residual branches, no layout copies
  BasicTransformerBlock :613-747                 -> LN -> fused QKV GEMM -> attention kernel -> GEMM(+res) ...
  MemoryEfficientIntraViewAttention :382-489     -> pnc_attn_views_f16, views are width strides
  MemoryEfficientInterViewAttentionTwo :493-610  -> same kernel, two KV segments (view 5: one, quirk Q1)
  CrossAttention / MemoryEfficientCrossAttention -> text keys: pnc_attn_views_f16 with 77 masked keys,
      :203-291 / :294-379                           K/V projected once per sample (not per pixel);
                                                    temporal self-attention: pnc_attn_temporal_f16
  FeedForward / GEGLU :91-117                    -> GEMM with fused GEGLU epilogue, GEMM(+res)
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .. import engine as E
from ..engine import Act, Packable, Runtime

HEAD_DIM = 64
INTRA_SEGS = [[0], [1], [2], [3], [4], [5]]
# MemoryEfficientInterViewAttentionTwo: K/V = [left, right]; view 0 = [5, 1]; for view 5 the reference's
# right-hand slice is empty (attention.py:549-551), so it attends to view 4 only.
INTER_SEGS = [[5, 1], [0, 2], [1, 3], [2, 4], [3, 5], [4]]


def exists(v):
    return v is not None


def default(v, d):
    return v if v is not None else (d() if callable(d) else d)


def Normalize(in_channels):
    """attention.py:129-132"""
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def panorama_grid(n_tokens: int, views: int = 6):
    """attention.py:428 — H = int(sqrt(N / 12)): six views of aspect 1:2 along the width.  `views` < 6: the tokens are a band
    of that many views of the panorama (engine.ViewShard)."""
    H = int(math.sqrt(n_tokens * 6 // views // 12))
    if H == 0 or n_tokens % H or (n_tokens // H) % views:
        raise ValueError(f"{n_tokens} tokens are not {views} views of aspect 1:2 per view")
    return H, n_tokens // H


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module, Packable):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("only the gated (GEGLU) feed-forward is on the Panacea path")
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.dim, self.inner_dim, self.dim_out = dim, inner_dim, dim_out
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))
        self._init_packable()

    def _pack(self):
        w1, b1 = E.pk_geglu(self.net[0].proj.weight, self.net[0].proj.bias)
        return dict(w1=w1, b1=b1, w2=E.pk_linear(self.net[2].weight), b2=E.pk_f32(self.net[2].bias))

    def _run(self, rt: Runtime, x16, M, res32, out32=None, out16=None, out16_lo=None):
        """out = FF(x16) + res32 -> out32 (may alias res32) and/or out16 (+ lo plane of a precise operand)."""
        pk = self.packed()
        hid = rt.empty((M, self.inner_dim), torch.float16)
        rt.be.gemm(x16, pk["w1"], M=M, N=2 * self.inner_dim, K=self.dim, lda=self.dim, bias=pk["b1"],
                   geglu=True, out16=hid, ldc16=self.inner_dim)
        rt.be.gemm(hid, pk["w2"], M=M, N=self.dim_out, K=self.inner_dim, lda=self.inner_dim, bias=pk["b2"],
                   res1=res32, ldr1=self.dim_out, out32=out32, ldc32=self.dim_out, out16=out16,
                   ldc16=self.dim_out, out16_lo=out16_lo)


def _ln_kwargs(rt: Runtime, ln, M: int, C: int):
    """kwargs that make a GEMM also write LayerNorm(out32 rows) as fp16 (PncGemmParams.ln_*): fused into the epilogue where
    a workgroup owns whole rows (level 0), the library's LayerNorm kernel right after the GEMM otherwise.  -> (kwargs, x16)"""
    if ln is None:
        return {}, None
    x16 = rt.empty((M, C), torch.float16)
    return dict(ln_gamma=ln[0], ln_beta=ln[1], ln_out16=x16, ldln=C, ln_eps=1e-5), x16


class _AttentionBase(nn.Module, Packable):
    """Parameters of every attention flavour: to_q / to_k / to_v (no bias) and to_out.0 (bias)."""
    kind = "plain"

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0, backend=None, **kwargs):
        super().__init__()
        if dim_head != HEAD_DIM:
            raise NotImplementedError(f"the gfx950 attention kernels are built for head dim {HEAD_DIM}, got {dim_head}")
        inner_dim = dim_head * heads
        self.is_self = context_dim is None
        context_dim = default(context_dim, query_dim)
        self.query_dim, self.context_dim, self.inner_dim = query_dim, context_dim, inner_dim
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self.backend = backend
        self._init_packable()

    def _pack(self):
        pk = dict(wo=E.pk_linear(self.to_out[0].weight), bo=E.pk_f32(self.to_out[0].bias))
        if self.is_self:
            pk["wqkv"] = E.pk_f16(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], dim=0))
        else:
            pk["wq"], pk["wk"], pk["wv"] = (E.pk_linear(self.to_q.weight), E.pk_linear(self.to_k.weight),
                                            E.pk_linear(self.to_v.weight))
        return pk

    # ---- text cross-attention: x16 [M, C] queries of F frames vs the 77 context tokens of each sample
    def _text_kv(self, rt: Runtime):
        """(k, ldk, vt, ldvt, vt_gstride) of this site.  The network projects the text keys/values of ALL its
        cross-attention sites of one width in two GEMMs up front (`project_text_kv`); a module used on its own
        projects its own."""
        hit = rt.text_kv.get(id(self))
        if hit is not None:
            return hit
        pk = self.packed()
        C, D, rows = self.inner_dim, rt.ctx16.shape[1], rt.B * E.TEXT_PAD
        k = rt.empty((rows, C), torch.float16)
        vt = rt.empty((rt.B, C, E.TEXT_PAD), torch.float16)
        rt.be.gemm(rt.ctx16, pk["wk"], M=rows, N=C, K=D, lda=D, out16=k, ldc16=C)
        rt.be.gemm(rt.ctx16, pk["wv"], M=rows, N=C, K=D, lda=D, out16t=vt, ldt=E.TEXT_PAD, t_rows=E.TEXT_PAD,
                   t_gstride=C * E.TEXT_PAD, n_split=0)
        return k, C, vt, E.TEXT_PAD, C * E.TEXT_PAD

    def _run_text(self, rt: Runtime, x16, F, H, W, res32, out32, ln=None):
        """`ln` = (gamma, beta) of the LayerNorm that follows the residual add: its fp16 output is returned"""
        pk = self.packed()
        C, M = self.inner_dim, F * H * W
        q = rt.empty((M, C), torch.float16)
        rt.be.gemm(x16, pk["wq"], M=M, N=C, K=self.query_dim, lda=self.query_dim, out16=q, ldc16=C)
        k, ldk, vt, ldvt, vt_gs = self._text_kv(rt)
        o = rt.empty((M, C), torch.float16)
        rt.be.attn_views(q, C, k, ldk, vt, ldvt, vt_gs, o, C, groups=F, heads=self.heads, H=H, W=W,
                         views=1, kvH=1, kvW=E.TEXT_PAD, kv_views=1, kv_rows_per_group=E.TEXT_PAD,
                         q_per_kv=F // rt.B, kv_valid=rt.n_text, segs=[[0]], scale=self.scale)
        lnkw, y16 = _ln_kwargs(rt, ln, M, self.query_dim)
        rt.be.gemm(o, pk["wo"], M=M, N=self.query_dim, K=C, lda=C, bias=pk["bo"], res1=res32,
                   ldr1=self.query_dim, out32=out32, ldc32=self.query_dim, **lnkw)
        return y16

    # ---- spatial self-attention over width-sliced views (views = 1: plain attention)
    def _run_views(self, rt: Runtime, x16, F, H, W, segs, res32, out32, ln=None):
        pk = self.packed()
        C, N = self.inner_dim, H * W
        M = F * N
        vs = rt.vshard if len(segs) == E.ViewShard.VIEWS else None     # `segs` speaks of the six views of the panorama
        views = vs.n_local if vs is not None else len(segs)
        if W % views:
            raise ValueError(f"grid width {W} is not divisible into {views} views")
        qk = rt.empty((M, 2 * C), torch.float16)
        vt = rt.empty((F, C, N), torch.float16)
        rt.be.gemm(x16, pk["wqkv"], M=M, N=3 * C, K=self.query_dim, lda=self.query_dim, out16=qk, ldc16=2 * C,
                   out16t=vt, ldt=N, t_rows=N, t_gstride=C * N, n_split=2 * C)
        o = rt.empty((M, C), torch.float16)
        Wv = W // views
        if vs is not None and any(u != v for v, row in enumerate(segs) for u in row):
            # this rank's band of views attends its neighbours' edge views too.  Round 5: the band's keys / values stay where the
            # QKV GEMM wrote them; the two neighbour views arrive in halo buffers of the band's geometry (PncAttnParams.k_halo) —
            # rounds 3-4 concatenated keys and values of n_local + 2 views (4.5 ms per evaluation at G = 1)
            k_halo, vt_halo = vs.halo_views(rt, qk.view(F, H, W, 2 * C)[..., C:], vt.view(F, C, H, W), M, 2 * C, C)
            rt.be.attn_views(qk, 2 * C, qk.view(-1)[C:], 2 * C, vt, N, C * N, o, C, groups=F, heads=self.heads,
                             H=H, W=W, views=views, kvH=H, kvW=W, kv_views=views, kv_rows_per_group=N, q_per_kv=1,
                             kv_valid=H * Wv, segs=vs.local_segments(segs, halo_ids=True), scale=self.scale,
                             k_halo=k_halo, vt_halo=vt_halo)
        else:
            if vs is not None:
                segs = [[i] for i in range(views)]
            rt.be.attn_views(qk, 2 * C, qk.view(-1)[C:], 2 * C, vt, N, C * N, o, C, groups=F, heads=self.heads,
                             H=H, W=W, views=views, kvH=H, kvW=W, kv_views=views, kv_rows_per_group=N, q_per_kv=1,
                             kv_valid=H * Wv, segs=segs, scale=self.scale)
        lnkw, y16 = _ln_kwargs(rt, ln, M, self.query_dim)
        rt.be.gemm(o, pk["wo"], M=M, N=self.query_dim, K=C, lda=C, bias=pk["bo"], res1=res32,
                   ldr1=self.query_dim, out32=out32, ldc32=self.query_dim, **lnkw)
        return y16

    # ---- temporal self-attention over the T frames of each pixel
    def _run_temporal(self, rt: Runtime, x16, N, res32, out32, ln=None):
        pk = self.packed()
        C = self.inner_dim
        M = rt.B * rt.T * N              # all T frames of N pixels per sample (N = pixels per rank when frame-sharded)
        qkv = rt.empty((M, 3 * C), torch.float16)
        rt.be.gemm(x16, pk["wqkv"], M=M, N=3 * C, K=self.query_dim, lda=self.query_dim, out16=qkv, ldc16=3 * C)
        o = rt.empty((M, C), torch.float16)
        flat = qkv.view(-1)
        rt.be.attn_temporal(flat, 3 * C, flat[C:], 3 * C, flat[2 * C:], 3 * C, o, C, B=rt.B, T=rt.T, Npix=N,
                            heads=self.heads, scale=self.scale)
        lnkw, y16 = _ln_kwargs(rt, ln, M, self.query_dim)
        rt.be.gemm(o, pk["wo"], M=M, N=self.query_dim, K=C, lda=C, bias=pk["bo"], res1=res32,
                   ldr1=self.query_dim, out32=out32, ldc32=self.query_dim, **lnkw)
        return y16


class CrossAttention(_AttentionBase):
    """attention.py:203-291"""
    kind = "plain"


class MemoryEfficientCrossAttention(_AttentionBase):
    """attention.py:294-379 (same parameters as CrossAttention; what the reference's torch 1.13 env runs)"""
    kind = "plain"


class MemoryEfficientIntraViewAttention(_AttentionBase):
    """attention.py:382-489"""
    kind = "intra-view"


class MemoryEfficientInterViewAttentionTwo(_AttentionBase):
    """attention.py:493-610"""
    kind = "inter-view"


class BasicTransformerBlock(nn.Module, Packable):
    """attention.py:613-747: x = attn1(LN1 x) + x ; x = attn2(LN2 x, text) + x ; x = ff(LN3 x) + x"""
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False, attn_mode="softmax", sdp_backend=None,
                 temporal_transformer_attn_type=None, spatial_only_attn_type=None):
        super().__init__()
        if attn_mode not in self.ATTENTION_MODES:
            raise ValueError(f"unknown attn_mode {attn_mode}")
        if disable_self_attn:
            raise NotImplementedError("disable_self_attn is not used on the Panacea path")
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = disable_self_attn
        self.temporal_transformer_attn_type = temporal_transformer_attn_type
        self.spatial_only_attn_type = spatial_only_attn_type
        if spatial_only_attn_type == "intra-view":
            a1 = MemoryEfficientIntraViewAttention
        elif spatial_only_attn_type == "inter-view":
            a1 = MemoryEfficientInterViewAttentionTwo
        else:
            a1 = attn_cls
        self.attn1 = a1(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=None,
                        backend=sdp_backend)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                              dropout=dropout, backend=sdp_backend)
        if self.attn2.is_self:
            raise NotImplementedError("attn2 without a text context is not on the Panacea path")
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.checkpoint = checkpoint
        self.dim = dim
        self._init_packable()

    def _pack(self):
        return {f"{n}{s}": E.pk_f32(getattr(getattr(self, n), "weight" if s == "w" else "bias"))
                for n in ("norm1", "norm2", "norm3") for s in ("w", "b")}

    def norm1_params(self):
        pk = self.packed()
        return pk["norm1w"], pk["norm1b"]

    def _run(self, rt: Runtime, t32, F, H, W, branch: str, last: bool, x16=None):
        """t32 [M, dim] fp32 stream, updated in place; returns the fp16 copy (hi, lo) of the final x when `last`
        (operand class `ff_out`).  `x16`: norm1(t32) when the GEMM that produced t32 has already written it.  norm2 / norm3
        are written by the residual GEMMs that precede them (PncGemmParams.ln_*), not by LayerNorm launches of their own."""
        pk = self.packed()
        C, N = self.dim, H * W
        M = F * N
        if x16 is None:
            x16 = E.layer_norm(rt, t32, M, C, pk["norm1w"], pk["norm1b"])
        ln2, ln3 = (pk["norm2w"], pk["norm2b"]), (pk["norm3w"], pk["norm3b"])
        nviews = rt.vshard.n_local if rt.vshard is not None else 6
        if branch == "temporal":
            x16 = self.attn1._run_temporal(rt, x16, N, t32, t32, ln=ln2)
        elif self.attn1.kind == "intra-view":
            ph, pw = panorama_grid(N, nviews)
            x16 = self.attn1._run_views(rt, x16, F, ph, pw, INTRA_SEGS, t32, t32, ln=ln2)
        elif self.attn1.kind == "inter-view":
            ph, pw = panorama_grid(N, nviews)
            x16 = self.attn1._run_views(rt, x16, F, ph, pw, INTER_SEGS, t32, t32, ln=ln2)
        else:
            if rt.vshard is not None:
                raise NotImplementedError("plain spatial self-attention spans the whole panorama; a view shard serves the "
                                          "intra-view / inter-view kinds")
            x16 = self.attn1._run_views(rt, x16, F, H, W, [[0]], t32, t32, ln=ln2)
        x16 = self.attn2._run_text(rt, x16, F, H, W, t32, t32, ln=ln3)
        out16 = out16lo = None
        if last:
            out16 = rt.empty((M, C), torch.float16)
            out16lo = rt.lo_plane((M, C), "ff_out")
        self.ff._run(rt, x16, M, t32, out32=None if last else t32, out16=out16, out16_lo=out16lo)
        return (out16, out16lo) if last else None


class SpatialTemporalTransformer(nn.Module, Packable):
    """attention.py:898-1134 — intra-view spatial -> cross-view -> cross-frame temporal branches."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None,
                 disable_self_attn=False, use_linear=False, attn_type="softmax", use_checkpoint=True,
                 sdp_backend=None, alpha=1, num_frames=4, temporal_transformer_attn_type=None,
                 spatial_only_attn_type=None, insert_crossview=False):
        super().__init__()
        if not use_linear:
            raise NotImplementedError("use_linear_in_transformer=False (1x1-conv projections) is not on the Panacea path")
        self.insert_crossview = insert_crossview
        self.num_frames = num_frames
        self.alpha = 1 if alpha == 1 else nn.Parameter(torch.rand(1, requires_grad=True))
        if exists(context_dim) and not isinstance(context_dim, (list, tuple)) and not _is_listconfig(context_dim):
            context_dim = [context_dim]
        if exists(context_dim):
            context_dim = list(context_dim)
            if depth != len(context_dim):
                assert all(c == context_dim[0] for c in context_dim), "need homogenous context_dim"
                context_dim = depth * [context_dim[0]]
        else:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        if inner_dim != in_channels:
            raise NotImplementedError("inner_dim != in_channels is not on the Panacea path")
        self.inner_dim = inner_dim
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.norm_temporal = Normalize(in_channels)
        self.proj_in_temporal = nn.Linear(in_channels, inner_dim)
        if insert_crossview:
            self.norm_crossview = Normalize(in_channels)
            self.proj_in_crossview = nn.Linear(in_channels, inner_dim)

        def blocks(**kw):
            return nn.ModuleList([
                BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                      disable_self_attn=disable_self_attn, attn_mode=attn_type,
                                      checkpoint=use_checkpoint, sdp_backend=sdp_backend, **kw)
                for d in range(depth)])
        self.transformer_blocks = blocks(spatial_only_attn_type=spatial_only_attn_type)
        self.transformer_blocks_temporal = blocks(temporal_transformer_attn_type=temporal_transformer_attn_type)
        if insert_crossview:
            assert spatial_only_attn_type == "intra-view"
            self.transformer_blocks_crossview = blocks(spatial_only_attn_type="inter-view")
        self.proj_out = zero_module(nn.Linear(inner_dim, in_channels))
        self.proj_out_temporal = zero_module(nn.Linear(inner_dim, in_channels))
        if insert_crossview:
            self.proj_out_crossview = zero_module(nn.Linear(inner_dim, in_channels))
        self.use_linear = use_linear
        self.identity_layer = nn.Identity()
        self._init_packable()

    def _pack(self):
        pk = {}
        sfx = ["", "_temporal"] + (["_crossview"] if self.insert_crossview else [])
        for s in sfx:
            n, pi, po = getattr(self, "norm" + s), getattr(self, "proj_in" + s), getattr(self, "proj_out" + s)
            pk["g" + s], pk["b" + s] = E.pk_f32(n.weight), E.pk_f32(n.bias)
            pk["wi" + s], pk["bi" + s] = E.pk_linear(pi.weight), E.pk_f32(pi.bias)
            pk["wo" + s], pk["bo" + s] = E.pk_linear(po.weight), E.pk_f32(po.bias)
        pk["pos"] = E.temporal_pos_table(self.num_frames, self.inner_dim).to(self.proj_in.weight.device)
        return pk

    def _branch(self, rt: Runtime, x: Act, sfx: str, blocks, branch: str, out16=None, out16_lo=None):
        pk = self.packed()
        C, M = x.C, x.M
        n16, n16lo = E.gn_spatial(rt, x.f32, x.F, x.N, C, pk["g" + sfx], pk["b" + sfx], 1e-6, False,
                                  split="gn_stt", part=x.gn_part)
        x.gn_part = None             # (the branches update x.f32 in place: the producer's records describe the first branch's input only)
        sh = rt.shard if branch == "temporal" else None
        if sh is not None:
            # Frame-sharded run: the temporal branch is pointwise per pixel (LN, projections, text cross-attention, FF) or
            # couples the T frames of ONE pixel (temporal self-attention), so it runs on all T frames of N/G pixels.
            # Exchanged: the GroupNorm output going in, the last block's fp16 output coming back (fp16 planes only).
            n16, n16lo = sh.to_pixels_planes([n16, n16lo], rt.B, x.N)       # (hi + lo plane in ONE exchange: round 6)
            Fb, Hb, Wb = rt.B * rt.T, 1, x.N // sh.G
        else:
            Fb, Hb, Wb = x.F, x.H, x.W
        Mb = Fb * Hb * Wb
        t32 = rt.empty((Mb, C), torch.float32)
        lnkw, x16 = _ln_kwargs(rt, blocks[0].norm1_params(), Mb, C)          # norm1 of the first block rides on proj_in
        if branch == "temporal":
            # + position table indexed by t = frame % T (attention.py:1117-1118)
            rt.be.gemm(n16, pk["wi" + sfx], M=Mb, N=C, K=C, lda=C, bias=pk["bi" + sfx], rowbias=pk["pos"],
                       rb_rows=Hb * Wb, rb_mod=rt.T, out32=t32, ldc32=C, a16_lo=n16lo, w_lo=E.wlo(pk, "wi" + sfx, n16lo), **lnkw)
        else:
            rt.be.gemm(n16, pk["wi" + sfx], M=Mb, N=C, K=C, lda=C, bias=pk["bi" + sfx], out32=t32, ldc32=C,
                       a16_lo=n16lo, w_lo=E.wlo(pk, "wi" + sfx, n16lo), **lnkw)
        p16 = p16lo = None
        for i, blk in enumerate(blocks):
            r = blk._run(rt, t32, Fb, Hb, Wb, branch, last=(i == len(blocks) - 1), x16=x16 if i == 0 else None)
            if r is not None:
                p16, p16lo = r
        if sh is not None:
            p16, p16lo = sh.to_frames_planes([p16, p16lo], rt.B, x.N)
        # x = proj_out(t) + x_in, in place on the stream
        rt.be.gemm(p16, pk["wo" + sfx], M=M, N=C, K=C, lda=C, bias=pk["bo" + sfx], res1=x.f32, ldr1=C,
                   out32=x.f32, ldc32=C, out16=out16, ldc16=C, a16_lo=p16lo, out16_lo=out16_lo,
                   w_lo=E.wlo(pk, "wo" + sfx, p16lo))

    def _run(self, rt: Runtime, x: Act, want_f16: bool = False) -> Act:
        if rt.T != self.num_frames:
            raise ValueError(f"runtime has {rt.T} frames per sample, module was built for {self.num_frames}")
        self._branch(rt, x, "", self.transformer_blocks, "spatial")
        if self.insert_crossview:
            self._branch(rt, x, "_crossview", self.transformer_blocks_crossview, "crossview")
        out16 = rt.empty((x.M, x.C), torch.float16) if want_f16 else None
        out16lo = rt.lo_plane((x.M, x.C), "stream", on=want_f16)
        self._branch(rt, x, "_temporal", self.transformer_blocks_temporal, "temporal", out16=out16, out16_lo=out16lo)
        return Act(x.F, x.H, x.W, x.C, f32=x.f32, f16=out16, f16_lo=out16lo)

    precision = "precise"      # operand policy of the reference-compatible entry below (the network sets rt.prec itself)

    def forward(self, x, context=None):
        """Reference-compatible entry: x (B*T, C, h, w) NCHW, context (B*T, n, D) already tiled over T."""
        from .util import act_from_nchw, runtime_for
        ctx = context[0] if isinstance(context, list) else context
        rt = runtime_for(x, self.num_frames)
        rt.prec = E.precision(self.precision)
        rt.set_context(ctx.view(rt.B, rt.T, *ctx.shape[1:])[:, 0])
        a = act_from_nchw(rt, x)
        return self._run(rt, a).to_nchw().to(x.dtype)


class TextKVProjector:
    """Batches the text K/V projections of every cross-attention site of a network.  They depend only on the (B, 77, D) context,
    so ALL sites share one GEMM (round 5; rounds 2-4: two per width, 16 launches of 20-80 workgroups per step): the weight rows
    are [W_k of every site | zero rows up to a multiple of 128 | W_v of every site], columns below `n_split` leave row-major (K_all,
    one column block per site), the others channel-major (V_all^T) — the QKV GEMM's output form.  138 launches of M = 160 rows
    become 2 per step (the reference instead re-projects them once per PIXEL in the temporal branch: attention.py:1122-1125).
    Every output element is the same dot product in the same K order as before: bit-identical."""

    def __init__(self, root: nn.Module):
        self.sites = [m.attn2 for m in root.modules() if isinstance(m, BasicTransformerBlock)]
        self._pk = None
        if not E.TEXTKV_ONE_GEMM:                      # A/B (PNC_TEXTKV_ONE_GEMM=0): one projector per width, as rounds 2-4 grouped
            by_c = {}
            for a in self.sites:
                by_c.setdefault(a.inner_dim, []).append(a)
            self.parts = [TextKVProjector.__new__(TextKVProjector) for _ in by_c]
            for p, sites in zip(self.parts, by_c.values()):
                p.sites, p._pk, p.parts = sites, None, None
        else:
            self.parts = None

    def pack(self):
        dims = {a.context_dim for a in self.sites}
        if len(dims) != 1:
            raise ValueError(f"cross-attention sites with different context widths {sorted(dims)}")
        wk = torch.cat([a.to_k.weight for a in self.sites], dim=0)
        wv = torch.cat([a.to_v.weight for a in self.sites], dim=0)
        nt = wk.shape[0]
        nkp = -(-nt // 128) * 128                      # PncGemmParams.n_split: a multiple of 128
        w = torch.cat([wk, wk.new_zeros(nkp - nt, wk.shape[1]), wv], dim=0)
        offs, o = [], 0
        for a in self.sites:
            offs.append(o)
            o += a.inner_dim
        return E.pk_f16(w), nt, nkp, offs, dims.pop()

    def run(self, rt: Runtime):
        if not self.sites:
            return
        if self.parts:
            for p in self.parts:
                p.run(rt)
            return
        sig = tuple((id(p), p._version, p.device) for a in self.sites for p in (a.to_k.weight, a.to_v.weight))      # (see EmbProjector.run)
        if self._pk is None or getattr(self, "_sig", None) != sig:
            with torch.no_grad():
                self._pk, self._sig = self.pack(), sig
        w, NT, NKp, offs, Dm = self._pk
        rows, D = rt.B * E.TEXT_PAD, rt.ctx16.shape[1]
        if Dm != D:
            raise ValueError(f"context width {D} does not match the cross-attention context_dim {Dm}")
        k = rt.empty((rows, NKp), torch.float16)
        vt = rt.empty((rt.B, NT, E.TEXT_PAD), torch.float16)
        rt.be.gemm(rt.ctx16, w, M=rows, N=NKp + NT, K=D, lda=D, out16=k, ldc16=NKp, out16t=vt, ldt=E.TEXT_PAD,
                   t_rows=E.TEXT_PAD, t_gstride=NT * E.TEXT_PAD, n_split=NKp)
        kf, vf = k.view(-1), vt.view(-1)
        for a, o in zip(self.sites, offs):
            rt.text_kv[id(a)] = (kf[o:], NKp, vf[o * E.TEXT_PAD:], E.TEXT_PAD, NT * E.TEXT_PAD)


def _is_listconfig(v) -> bool:
    return type(v).__name__ == "ListConfig"


def create_1d_absolute_sin_cos_embedding(pos_len, dim):
    """attention.py:1140-1159 as it evaluates (degenerate table)."""
    assert dim % 2 == 0, "wrong dimension!"
    return E.temporal_pos_table(pos_len, dim)
