"""Kernel chains of the C-ABI against the ORACLE's own primitives (oracle/panacea_oracle.py: pinned on the reference by
tests/golden) — not against tests/emu.py, the builder's emulation of the same C-ABI (a bug shared by a kernel and
its emulation would only be caught one level up).  Each test drives the launches a module issues for one reference call-site with
fp16-representable weights and compares with the oracle function of that call-site evaluated in fp32 on the same tensors:

    _view_attention (intra / inter)   QKV GEMM (+ channel-major V^T) -> pnc_attn_views_f16 -> out-projection GEMM
    cross_attention (77 text tokens)  q / K / V^T GEMMs -> pnc_attn_views_f16 (masked keys) -> out-projection GEMM
    cross_attention (temporal self)   QKV GEMM -> pnc_attn_temporal_f16 -> out-projection GEMM
    feed_forward                      GEGLU GEMM -> GEMM
    _temporal_conv                    pnc_groupnorm_temporal_silu -> conv1d GEMM + residual
    _conv(silu(_gn(x)))               pnc_groupnorm_stats / apply -> conv3x3 GEMM
    _ln, timestep_embedding           pnc_layernorm, pnc_timestep_embedding

Tolerances: fp16 operands with fp32 accumulation against fp32 — max-abs 4e-3 x the output scale (measured 1-2e-3)."""
import math

import pytest
import torch

from oracle import panacea_oracle as po
from panacea_amd import engine as E, hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def r16(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().float()


def close(name, got, ref, tol=4e-3):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    scale = max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    print(f"{name}: max-abs {err:.3e} (ref max {ref.abs().max().item():.2f})")
    assert torch.isfinite(got).all() and err <= tol * scale, (name, err, scale)


def _attn_sd(C, ctx_dim=None, seed=0):
    cd = ctx_dim or C
    return {"a.to_q.weight": r16(C, C, scale=C ** -0.5, seed=seed), "a.to_k.weight": r16(C, cd, scale=cd ** -0.5, seed=seed + 1),
            "a.to_v.weight": r16(C, cd, scale=cd ** -0.5, seed=seed + 2), "a.to_out.0.weight": r16(C, C, scale=C ** -0.5, seed=seed + 3),
            "a.to_out.0.bias": r16(C, scale=0.1, seed=seed + 4)}


@pytest.mark.parametrize("inter", [False, True])
@pytest.mark.parametrize("b,H,C", [(2, 8, 128), (1, 16, 320)])
def test_view_attention_chain_vs_oracle(inter, b, H, C):
    """attention.py:407-489 / 518-610 incl. the view-5 quirk, through the oracle's _view_attention"""
    from panacea_amd.nn.attention import INTER_SEGS, INTRA_SEGS
    heads, W = C // 64, 12 * H
    N = H * W
    sd = _attn_sd(C, seed=10 + H)
    x = r16(b, N, C, seed=3)
    ref = po._view_attention(sd, "a", x, heads, inter)
    dev = lambda t: t.to(DEV)            # noqa: E731
    x16 = dev(x).half().view(b * N, C)
    wqkv = E.pk_f16(torch.cat([sd["a.to_q.weight"], sd["a.to_k.weight"], sd["a.to_v.weight"]], 0)).to(DEV)
    qk = torch.empty(b * N, 2 * C, device=DEV, dtype=torch.float16)
    vt = torch.empty(b, C, N, device=DEV, dtype=torch.float16)
    hip.gemm(x16, wqkv, M=b * N, N=3 * C, K=C, lda=C, out16=qk, ldc16=2 * C, out16t=vt, ldt=N, t_rows=N, t_gstride=C * N, n_split=2 * C)
    o = torch.empty(b * N, C, device=DEV, dtype=torch.float16)
    hip.attn_views(qk, 2 * C, qk.view(-1)[C:], 2 * C, vt, N, C * N, o, C, groups=b, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W,
                   kv_views=6, kv_rows_per_group=N, q_per_kv=1, kv_valid=H * (W // 6), segs=INTER_SEGS if inter else INTRA_SEGS,
                   scale=64 ** -0.5)
    out = torch.empty(b * N, C, device=DEV)
    hip.gemm(o, E.pk_linear(sd["a.to_out.0.weight"]).to(DEV), M=b * N, N=C, K=C, lda=C, bias=dev(sd["a.to_out.0.bias"]), out32=out, ldc32=C)
    torch.cuda.synchronize()
    close(f"view attention inter={inter} C={C}", out.view(b, N, C), ref)


def test_text_cross_attention_chain_vs_oracle():
    """attention.py:229-291 with the 77 context tokens (keys padded to 80 rows, masked in the kernel)"""
    b, T, N, C, D = 2, 2, 192, 128, 64
    heads = C // 64
    sd = _attn_sd(C, ctx_dim=D, seed=30)
    x = r16(b * T, N, C, seed=5)
    ctx = r16(b, 77, D, seed=6)
    ref = po.cross_attention(sd, "a", x, ctx.repeat_interleave(T, 0), heads)
    dev = lambda t: t.to(DEV)            # noqa: E731
    M = b * T * N
    x16 = dev(x).half().view(M, C)
    c16 = torch.zeros(b * 80, D, device=DEV, dtype=torch.float16)
    c16.view(b, 80, D)[:, :77] = dev(ctx).half()
    q = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.gemm(x16, E.pk_linear(sd["a.to_q.weight"]).to(DEV), M=M, N=C, K=C, lda=C, out16=q, ldc16=C)
    k = torch.empty(b * 80, C, device=DEV, dtype=torch.float16)
    hip.gemm(c16, E.pk_linear(sd["a.to_k.weight"]).to(DEV), M=b * 80, N=C, K=D, lda=D, out16=k, ldc16=C)
    v = torch.empty(b * 80, C, device=DEV, dtype=torch.float16)
    hip.gemm(c16, E.pk_linear(sd["a.to_v.weight"]).to(DEV), M=b * 80, N=C, K=D, lda=D, out16=v, ldc16=C)
    vt = v.view(b, 80, C).permute(0, 2, 1).contiguous()
    o = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.attn_views(q, C, k, C, vt, 80, C * 80, o, C, groups=b * T, heads=heads, H=1, W=N, views=1, kvH=1, kvW=80, kv_views=1,
                   kv_rows_per_group=80, q_per_kv=T, kv_valid=77, segs=[[0]], scale=64 ** -0.5)
    out = torch.empty(M, C, device=DEV)
    hip.gemm(o, E.pk_linear(sd["a.to_out.0.weight"]).to(DEV), M=M, N=C, K=C, lda=C, bias=dev(sd["a.to_out.0.bias"]), out32=out, ldc32=C)
    torch.cuda.synchronize()
    close("text cross-attention", out.view(b * T, N, C), ref)


def test_temporal_self_attention_chain_vs_oracle():
    """CrossAttention as self-attention over the T frames of a pixel ("(b h w) t c", attention.py:1106-1134)"""
    B, T, Npix, C = 2, 4, 96, 128
    heads = C // 64
    sd = _attn_sd(C, seed=40)
    x = r16(B, T, Npix, C, seed=7)                                   # resident layout: row (b, t, p)
    ref = po.cross_attention(sd, "a", x.permute(0, 2, 1, 3).reshape(B * Npix, T, C), None, heads)      # (b p) t c
    ref = ref.view(B, Npix, T, C).permute(0, 2, 1, 3)
    M = B * T * Npix
    x16 = x.to(DEV).half().view(M, C)
    wqkv = E.pk_f16(torch.cat([sd["a.to_q.weight"], sd["a.to_k.weight"], sd["a.to_v.weight"]], 0)).to(DEV)
    qkv = torch.empty(M, 3 * C, device=DEV, dtype=torch.float16)
    hip.gemm(x16, wqkv, M=M, N=3 * C, K=C, lda=C, out16=qkv, ldc16=3 * C)
    o = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.attn_temporal(qkv, 3 * C, qkv[:, C:], 3 * C, qkv[:, 2 * C:], 3 * C, o, C, B=B, T=T, Npix=Npix, heads=heads, scale=64 ** -0.5)
    out = torch.empty(M, C, device=DEV)
    hip.gemm(o, E.pk_linear(sd["a.to_out.0.weight"]).to(DEV), M=M, N=C, K=C, lda=C, bias=sd["a.to_out.0.bias"].to(DEV), out32=out, ldc32=C)
    torch.cuda.synchronize()
    close("temporal self-attention", out.view(B, T, Npix, C), ref)


@pytest.mark.parametrize("M,C", [(768, 128), (4096, 320)])
def test_feed_forward_chain_vs_oracle(M, C):
    """FeedForward / GEGLU (attention.py:91-117: gate = second half, erf GELU) + the block's residual"""
    sd = {"f.net.0.proj.weight": r16(8 * C, C, scale=C ** -0.5, seed=50), "f.net.0.proj.bias": r16(8 * C, scale=0.1, seed=51),
          "f.net.2.weight": r16(C, 4 * C, scale=(4 * C) ** -0.5, seed=52), "f.net.2.bias": r16(C, scale=0.1, seed=53)}
    x = r16(M, C, seed=54)
    res = r16(M, C, seed=55)
    ref = po.feed_forward(sd, "f", x) + res
    w1, b1 = E.pk_geglu(sd["f.net.0.proj.weight"], sd["f.net.0.proj.bias"])
    hid = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
    hip.gemm(x.to(DEV).half(), w1.to(DEV), M=M, N=8 * C, K=C, lda=C, bias=b1.to(DEV), geglu=True, out16=hid, ldc16=4 * C)
    out = res.to(DEV).clone()
    hip.gemm(hid, E.pk_linear(sd["f.net.2.weight"]).to(DEV), M=M, N=C, K=4 * C, lda=4 * C, bias=sd["f.net.2.bias"].to(DEV),
             res1=out, ldr1=C, out32=out, ldc32=C)
    torch.cuda.synchronize()
    close(f"feed-forward C={C}", out, ref)


@pytest.mark.parametrize("B,T,H,W,C", [(2, 4, 4, 12, 64), (1, 8, 8, 24, 320)])
def test_temporal_conv_site_vs_oracle(B, T, H, W, C):
    """`h + conv1d(SiLU(GN(h)))` on "(b h w) c t" (openaimodel.py:505-515): temporal GroupNorm kernel + conv1d GEMM"""
    sd = {"t.0.weight": r16(C, scale=0.3, seed=60) + 1.0, "t.0.bias": r16(C, scale=0.2, seed=61),
          "t.2.weight": r16(C, C, 3, scale=(3 * C) ** -0.5, seed=62), "t.2.bias": r16(C, scale=0.1, seed=63)}
    h = r16(B * T, C, H, W, seed=64)
    ref = po._temporal_conv(sd, "t", h, T)
    Npix, M = H * W, B * T * H * W
    h32 = h.to(DEV).permute(0, 2, 3, 1).reshape(M, C).contiguous()
    t16 = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.groupnorm_temporal_silu(h32, B, T, Npix, C, sd["t.0.weight"].to(DEV), sd["t.0.bias"].to(DEV), 1e-5, t16)
    out = h32.clone()
    hip.gemm(t16, E.pk_conv1d(sd["t.2.weight"]).to(DEV), M=M, N=C, K=3 * C, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix),
             bias=sd["t.2.bias"].to(DEV), res1=out, ldr1=C, out32=out, ldc32=C)
    torch.cuda.synchronize()
    close(f"temporal conv site C={C}", out.view(B * T, H, W, C).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("F,H,W,Cin,Cout", [(2, 8, 24, 64, 128), (1, 16, 48, 320, 320)])
def test_groupnorm_silu_conv3x3_vs_oracle(F, H, W, Cin, Cout):
    """in_layers of a ResBlock3D (openaimodel.py:409-413): GroupNorm(32, eps 1e-5) + SiLU + 3x3 conv"""
    sd = {"n.weight": r16(Cin, scale=0.3, seed=70) + 1.0, "n.bias": r16(Cin, scale=0.2, seed=71),
          "c.weight": r16(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=72), "c.bias": r16(Cout, scale=0.1, seed=73)}
    x = r16(F, Cin, H, W, seed=74) * 1.5 + 0.3
    ref = po._conv(sd, "c", torch.nn.functional.silu(po._gn(sd, "n", x, 1e-5)))
    N, M = H * W, F * H * W
    x32 = x.to(DEV).permute(0, 2, 3, 1).reshape(M, Cin).contiguous()
    ppc = E._ppc(N)
    nchunk = (N + ppc - 1) // ppc
    part = torch.empty(F * nchunk * 96, device=DEV)
    a16 = torch.empty(M, Cin, device=DEV, dtype=torch.float16)
    hip.groupnorm_stats(x32, Cin, F, N, Cin, ppc, part)
    hip.groupnorm_apply(x32, Cin, F, N, Cin, ppc, part, sd["n.weight"].to(DEV), sd["n.bias"].to(DEV), 1e-5, True, a16, Cin)
    out = torch.empty(M, Cout, device=DEV)
    hip.gemm(a16, E.pk_conv3x3(sd["c.weight"]).to(DEV), M=M, N=Cout, K=9 * Cin, a_mode=hip.A_CONV3X3,
             conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0), bias=sd["c.bias"].to(DEV), out32=out, ldc32=Cout)
    torch.cuda.synchronize()
    close(f"GN+SiLU+conv3x3 {Cin}->{Cout}", out.view(F, H, W, Cout).permute(0, 3, 1, 2), ref)


def test_layernorm_and_timestep_embedding_vs_oracle():
    C, M = 640, 300
    sd = {"l.weight": r16(C, scale=0.3, seed=80) + 1.0, "l.bias": r16(C, scale=0.2, seed=81)}
    x = r16(M, C, seed=82) * 2.0 - 0.5
    y = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.layernorm(x.to(DEV), C, M, C, sd["l.weight"].to(DEV), sd["l.bias"].to(DEV), 1e-5, y, C)
    close("layernorm", y, po._ln(sd, "l", x), 2e-3)
    t = torch.tensor([999, 500, 39, 0, 1, 777], dtype=torch.int64)
    out = torch.empty(t.numel(), 320, device=DEV)
    hip.timestep_embedding(t.to(DEV), t.numel(), 320, E.timestep_freqs(320, torch.device(DEV)), out)
    torch.cuda.synchronize()
    close("timestep embedding", out, po.timestep_embedding(t, 320), 2e-4)


# ---- round 6: more call-sites compared with the oracle's primitives / plain torch fp32 directly ----------------
@pytest.mark.parametrize("F,H,W,C", [(4, 32, 64, 320), (2, 16, 96, 640)])
def test_stencil_tile_conv3x3_vs_oracle(F, H, W, C):
    """the 3x3 conv over spatial tiles that hold their halo (gemm_stencil_tile.hip: the ResBlock3D convs of the full-size network),
    forced on a grid the size heuristic would hand to the per-tap gather, against the oracle's _conv — incl. residual + fp16 output"""
    sd = {"c.weight": r16(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=90), "c.bias": r16(C, scale=0.1, seed=91)}
    x = r16(F, C, H, W, seed=92)
    res = r16(F, C, H, W, seed=93)
    ref = po._conv(sd, "c", x) + res
    M = F * H * W
    x16 = x.to(DEV).permute(0, 2, 3, 1).reshape(M, C).contiguous().half()
    o32 = res.to(DEV).permute(0, 2, 3, 1).reshape(M, C).contiguous()
    o16 = torch.empty(M, C, device=DEV, dtype=torch.float16)
    prev = hip.set_option(hip.OPT_STENCIL_TILES, 2)
    try:
        hip.gemm(x16, E.pk_conv3x3(sd["c.weight"]).to(DEV), M=M, N=C, K=9 * C, a_mode=hip.A_CONV3X3,
                 conv=dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0), bias=sd["c.bias"].to(DEV),
                 res1=o32, ldr1=C, out32=o32, ldc32=C, out16=o16, ldc16=C)
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_STENCIL_TILES, prev)
    close(f"stencil conv3x3 {C} + res", o32.view(F, H, W, C).permute(0, 3, 1, 2), ref)
    close(f"stencil conv3x3 {C} fp16 out", o16.view(F, H, W, C).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("mode", ["down", "up"])
def test_down_and_upsample_convs_vs_torch(mode):
    """Downsample (openaimodel.py:161-201: 3x3 stride 2, pad 1) and Upsample (:106-142: nearest x2, then 3x3) with the stride / the
    nearest-neighbour expansion folded into the conv's A gather — no strided or upsampled tensor exists — against torch fp32"""
    import torch.nn.functional as TF
    F, H, W, C = 2, 16, 96, 320
    w = r16(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=95)
    b = r16(C, scale=0.1, seed=96)
    x = r16(F, C, H, W, seed=97)
    if mode == "down":
        ref = TF.conv2d(x, w, b, stride=2, padding=1)
        conv = dict(Cin=C, Hin=H, Win=W, Hout=H // 2, Wout=W // 2, stride=2, upsample=0)
    else:
        ref = TF.conv2d(TF.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
        conv = dict(Cin=C, Hin=H, Win=W, Hout=2 * H, Wout=2 * W, stride=1, upsample=1)
    M = F * conv["Hout"] * conv["Wout"]
    x16 = x.to(DEV).permute(0, 2, 3, 1).reshape(-1, C).contiguous().half()
    out = torch.empty(M, C, device=DEV)
    hip.gemm(x16, E.pk_conv3x3(w).to(DEV), M=M, N=C, K=9 * C, a_mode=hip.A_CONV3X3, conv=conv, bias=b.to(DEV), out32=out, ldc32=C)
    torch.cuda.synchronize()
    close(f"{mode}sample conv", out.view(F, conv["Hout"], conv["Wout"], C).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("M,N,K", [(4096, 320, 320), (2048, 640, 1280)])
def test_precise_operand_gemm_vs_fp64(M, N, K):
    """a split operand (fp16 hi + e4m3 lo plane, lo pass on the scaled fp8 MFMA against the e4m3 weight copy: DESIGN.md section 4) against
    the fp64 product of the UNROUNDED fp32 activations: the pair must resolve the operand far below fp16's 2^-11 — measured ~30x
    closer than the plain fp16 operand on the same data"""
    g = torch.Generator().manual_seed(5)
    a32 = torch.randn(M, K, generator=g) * 1.5
    w = r16(N, K, scale=K ** -0.5, seed=6)
    ref = (a32.double() @ w.double().t()).float()
    a = a32.to(DEV)
    hi = torch.empty(M, K, device=DEV, dtype=torch.float16)
    lo = torch.empty(M, K, device=DEV, dtype=torch.uint8)
    hip.cast_f16(a, M * K, hi, lo)
    w16 = w.half().to(DEV)
    out_p, out_f = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    hip.gemm(hi, w16, M=M, N=N, K=K, lda=K, out32=out_p, ldc32=N, a16_lo=lo, w_lo=E.pk_lo8(w16))
    hip.gemm(hi, w16, M=M, N=N, K=K, lda=K, out32=out_f, ldc32=N)
    torch.cuda.synchronize()
    ep, ef = (out_p.cpu() - ref).abs().max().item(), (out_f.cpu() - ref).abs().max().item()
    print(f"precise operand GEMM {M}x{N}x{K}: max-abs err split {ep:.2e}, plain fp16 operand {ef:.2e}")
    assert ep <= 2.5e-4 * max(1.0, ref.abs().max().item()) and ep * 8 <= ef, (ep, ef)


def test_concat_add_with_groupnorm_records_vs_torch():
    """th.cat([h, hs.pop() + control.pop()], dim=1) (controlmodel.py:196-197) in one pass that also writes the GroupNorm(32) records
    of the result: values against torch, the records through pnc_groupnorm_apply against torch's group_norm"""
    import torch.nn.functional as TF
    F, H, W, C1, C2 = 2, 16, 96, 640, 320
    N, M, C = H * W, F * H * W, C1 + C2
    h, s, c = r16(M, C1, seed=101) * 1.3, r16(M, C2, seed=102), r16(M, C2, seed=103) * 0.5
    ref = torch.cat([h, s + c], dim=1)
    ppc = E._ppc(N)
    nrec = -(-N // ppc)
    o32 = torch.empty(M, C, device=DEV)
    o16 = torch.empty(M, C, device=DEV, dtype=torch.float16)
    part = torch.empty(F * nrec * 96, device=DEV)
    hip.concat_add(h.to(DEV), C1, s.to(DEV), c.to(DEV), C2, M, o32, o16, None, gn_part=part, frames=F, ppc=ppc)
    gamma, beta = (r16(C, scale=0.3, seed=104) + 1.0), r16(C, scale=0.2, seed=105)
    y = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hip.groupnorm_apply(o32, C, F, N, C, ppc, part, gamma.to(DEV), beta.to(DEV), 1e-5, False, y, C, n_records=nrec)
    torch.cuda.synchronize()
    assert torch.equal(o32.cpu(), ref)                                   # fp32 adds: exact
    close("concat fp16 plane", o16, ref, 1e-3)
    gn = TF.group_norm(ref.view(F, N, C).permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1).reshape(M, C)
    close("GroupNorm from the concat's records", y, gn, 2e-3)
