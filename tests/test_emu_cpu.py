"""The torch emulation of the C-ABI (tests/emu.py) is itself checked against independent torch
module-level references, so that the GPU kernel tests compare against a trusted checker."""
import torch
import torch.nn.functional as TF

import emu


def test_emu_conv3x3_modes():
    torch.manual_seed(0)
    F, Hin, Win, Cin, N = 2, 6, 10, 8, 16
    x = torch.randn(F, Cin, Hin, Win).half()
    w = torch.randn(N, Cin, 3, 3).half()
    xt = x.permute(0, 2, 3, 1).contiguous()
    wp = w.permute(0, 2, 3, 1).reshape(N, 9 * Cin).contiguous()
    for stride, up in [(1, 0), (2, 0), (1, 1)]:
        xin = TF.interpolate(x.float(), scale_factor=2, mode="nearest") if up else x.float()
        ref = TF.conv2d(xin, w.float(), stride=stride, padding=1)
        Hout, Wout = ref.shape[2:]
        out = torch.zeros(F * Hout * Wout, N)
        emu.gemm(xt, wp, M=F * Hout * Wout, N=N, K=9 * Cin, a_mode=emu.A_CONV3X3,
                 conv=dict(Cin=Cin, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride, upsample=up),
                 out32=out, ldc32=N)
        assert torch.allclose(out.view(F, Hout, Wout, N).permute(0, 3, 1, 2), ref, atol=1e-4)


def test_emu_conv3x3_chunked_k_order_matches_packer():
    from panacea_amd import engine as E
    torch.manual_seed(5)
    F, H, W, Cin, N = 1, 5, 7, 128, 32
    x = torch.randn(F, Cin, H, W).half()
    w = torch.randn(N, Cin, 3, 3).half() * 0.05
    ref = TF.conv2d(x.float(), w.float(), padding=1)
    out = torch.zeros(F * H * W, N)
    emu.gemm(x.permute(0, 2, 3, 1).contiguous(), E.pk_conv3x3(w), M=F * H * W, N=N, K=9 * Cin, a_mode=emu.A_CONV3X3,
             conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0), out32=out, ldc32=N)
    assert torch.allclose(out.view(F, H, W, N).permute(0, 3, 1, 2), ref, atol=2e-3)


def test_emu_conv1d_temporal_and_gn():
    torch.manual_seed(1)
    B, T, Npix, C = 2, 4, 5, 64
    x = torch.randn(B * T, C, Npix, 1).half()                 # (b t) c h w
    w = torch.randn(C, C, 3).half()
    xr = x.float().view(B, T, C, Npix).permute(0, 3, 2, 1).reshape(B * Npix, C, T)   # (b h w) c t
    ref = TF.conv1d(xr, w.float(), padding=1).view(B, Npix, C, T).permute(0, 3, 1, 2).reshape(B * T * Npix, C)
    tok = x.view(B * T, C, Npix).permute(0, 2, 1).reshape(B * T * Npix, C).contiguous()
    wp = w.permute(0, 2, 1).reshape(C, 3 * C).contiguous()
    out = torch.zeros(B * T * Npix, C)
    emu.gemm(tok, wp, M=B * T * Npix, N=C, K=3 * C, a_mode=emu.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix),
             out32=out, ldc32=C)
    assert torch.allclose(out, ref, atol=1e-4)
    g, b = torch.randn(C), torch.randn(C)
    y = torch.zeros(B * T * Npix, C, dtype=torch.float16)
    emu.groupnorm_temporal_silu(tok.float(), B, T, Npix, C, g, b, 1e-5, y)
    refn = TF.silu(TF.group_norm(xr, 32, g, b, 1e-5)).view(B, Npix, C, T).permute(0, 3, 1, 2).reshape(-1, C)
    assert torch.allclose(y.float(), refn, atol=2e-3)


def test_emu_groupnorm_spatial_chunks():
    torch.manual_seed(2)
    F, Npix, C, ppc = 2, 50, 64, 16
    x = torch.randn(F * Npix, C) * 2 + 1
    g, b = torch.randn(C), torch.randn(C)
    nchunk = (Npix + ppc - 1) // ppc
    part = torch.zeros(F * nchunk * 32 * 3)
    y = torch.zeros(F * Npix, C, dtype=torch.float16)
    emu.groupnorm_stats(x, C, F, Npix, C, ppc, part)
    emu.groupnorm_apply(x, C, F, Npix, C, ppc, part, g, b, 1e-6, 0, y, C)
    ref = TF.group_norm(x.view(F, Npix, C).permute(0, 2, 1), 32, g, b, 1e-6).permute(0, 2, 1).reshape(-1, C)
    assert torch.allclose(y.float(), ref, atol=3e-3)


def test_emu_attn_views_matches_per_view_loop():
    torch.manual_seed(3)
    G, H, W, heads = 2, 4, 48, 2
    C, N, Wv = heads * 64, H * W, W // 6
    q, k, v = (torch.randn(G * N, C).half() for _ in range(3))
    vt = v.view(G, N, C).permute(0, 2, 1).contiguous()
    segs = [[5, 1], [0, 2], [1, 3], [2, 4], [3, 5], [4]]
    o = torch.zeros(G * N, C, dtype=torch.float16)
    emu.attn_views(q, C, k, C, vt, N, C * N, o, C, groups=G, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W,
                   kv_views=6, kv_rows_per_group=N, q_per_kv=1, kv_valid=H * Wv, segs=segs, scale=0.125)
    qg, kg, vg = (t.float().view(G, H, W, heads, 64) for t in (q, k, v))
    for view in range(6):
        qq = qg[:, :, view * Wv:(view + 1) * Wv].reshape(G, -1, heads, 64).transpose(1, 2)
        kk = torch.cat([kg[:, :, u * Wv:(u + 1) * Wv].reshape(G, -1, heads, 64) for u in segs[view]], 1).transpose(1, 2)
        vv = torch.cat([vg[:, :, u * Wv:(u + 1) * Wv].reshape(G, -1, heads, 64) for u in segs[view]], 1).transpose(1, 2)
        ref = TF.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(G, H, Wv, C)
        got = o.view(G, H, W, C)[:, :, view * Wv:(view + 1) * Wv].float()
        assert torch.allclose(got, ref, atol=2e-3)


def test_emu_geglu_and_transposed_split():
    torch.manual_seed(4)
    M, C = 64, 64
    a = torch.randn(M, C).half()
    w = torch.randn(8 * C, C).half() * 0.1
    bias = torch.randn(8 * C)
    # interleave 32 value rows / 32 gate rows as the engine's packer does
    val, gate = w[: 4 * C], w[4 * C:]
    wi = torch.stack([val.view(-1, 32, C), gate.view(-1, 32, C)], 1).reshape(8 * C, C).contiguous()
    bi = torch.stack([bias[: 4 * C].view(-1, 32), bias[4 * C:].view(-1, 32)], 1).reshape(-1).contiguous()
    out = torch.zeros(M, 4 * C, dtype=torch.float16)
    emu.gemm(a, wi, M=M, N=8 * C, K=C, lda=C, bias=bi, geglu=True, out16=out, ldc16=4 * C)
    proj = a.float() @ w.float().t() + bias
    ref = proj[:, : 4 * C] * TF.gelu(proj[:, 4 * C:])
    assert torch.allclose(out.float(), ref, atol=5e-3, rtol=5e-3)
    qk = torch.zeros(M, 128, dtype=torch.float16)
    vt = torch.zeros(2, 64, 40, dtype=torch.float16)
    w3 = torch.randn(192, C).half() * 0.1
    emu.gemm(a, w3, M=M, N=192, K=C, lda=C, out16=qk, ldc16=128, out16t=vt, ldt=40, t_rows=32,
             t_gstride=64 * 40, n_split=128)
    full = (a.float() @ w3.float().t()).half()
    assert torch.equal(qk, full[:, :128])
    assert torch.equal(vt[:, :, :32], full[:, 128:].view(2, 32, 64).permute(0, 2, 1))


def test_mx8_lo_plane_format_reference():
    """The MX fp8 lo plane: block maxima land in [128, 256) of e4m3, the round trip is exact for values
    e4m3 holds, relative error of any element <= 2^-4 of its block maximum... and a split operand carrying it is ~2^-15 accurate:
    8x closer to the fp32 value than the plain fp16 rounding, which is what the error budget needs (tools/exp/error_budget.py)."""
    import torch
    g = torch.Generator().manual_seed(5)
    v = torch.randn(64, 320, generator=g) * torch.logspace(-3, 2, 320)[None, :]            # wide per-channel range
    hi = v.half()
    q, sc = emu._lo_mx8(v, hi)
    assert q.dtype == torch.uint8 and sc.dtype == torch.uint8 and sc.shape == (64, 10)
    r = (v - hi.float()) * emu.LO_SCALE
    back = emu.mx8_dequant(q, sc)
    blk = r.reshape(64, 10, 32).abs().amax(-1, keepdim=True)
    assert ((back - r).reshape(64, 10, 32).abs() <= blk * 2.0 ** -4 + 1e-30).all()
    scaled_max = blk / torch.exp2(sc.float().reshape(64, 10, 1) - 127.0)
    assert ((scaled_max >= 128.0) & (scaled_max < 256.0) | (blk == 0)).all()
    err16 = (hi.float() - v).abs()
    err_mx = (hi.float() + back / emu.LO_SCALE - v).abs()
    assert err_mx.max() <= err16.max() / 8.0
    # exactly representable values survive: integers up to 16 times a power of two
    x = (torch.arange(-16, 16).float() * 0.25).repeat(2, 1)
    qq, ss = emu.mx8_quant(x)
    assert torch.equal(emu.mx8_dequant(qq, ss), x)
    z, zs = emu.mx8_quant(torch.zeros(1, 32))
    assert z.eq(0).all() and zs.item() == 0


