"""e4m3 lo planes of precise operands (PNC_LO_E4M3, include/panacea_hip.h) on the MI355X: every producer against the
emulation's bytes, and the lo pass of the GEMM family (block-scaled fp8 MFMA, e4m3 copy of the weights) against the
emulation and against the fp64 product of the unsplit operands."""
import pytest
import torch

import emu
from panacea_amd import engine, hip
from test_kernels_gpu import DEV, TILES, check, rnd, _run_both, _splits

pytestmark = pytest.mark.gpu


def _deq(b):
    return b.view(torch.float8_e4m3fn).float()


def _split8(v):
    hi = v.half()
    return hi, emu._lo(v, hi, torch.empty(0, dtype=torch.uint8))


def _same_bytes(name, got, ref, min_equal=1.0):
    """e4m3 planes: bit equality (min_equal = 1) or, where the fp32 value behind the byte may differ by an ulp between kernel
    and emulation, at least `min_equal` of the bytes equal and the rest one e4m3 step apart"""
    assert got.dtype == torch.uint8 and ref.dtype == torch.uint8
    eq = (got == ref).float().mean().item()
    assert eq >= min_equal, f"{name}: only {eq:.5f} of the e4m3 bytes equal"
    if eq < 1.0:
        d = (_deq(got) - _deq(ref)).abs()
        assert (d <= 0.126 * _deq(ref).abs() + 2.0 ** -9).all(), f"{name}: bytes more than one e4m3 step apart"


def test_e4m3_lo_planes_of_every_producer():
    def rec(hi, lo8):
        return hi.float() + _deq(lo8) / 2048.0
    # exact fp32 sources: the lo plane must equal the emulation's byte for byte, and hi + lo must be ~2^-15 of v
    Mh, Ca, Cb = 300, 128, 64
    aa, ss, cc = rnd(Mh, Ca) * 5, rnd(Mh, Cb), rnd(Mh, Cb)
    aa[0, :8] = torch.tensor([0.0, 1e-6, -3e-5, 700.0, -1200.0, 60000.0, 2.0 ** -14, 1.0], device=DEV)   # zeros, tiny, lo beyond 448
    for fn in ("cast", "add", "concat"):
        if fn == "concat":
            h16, hlo, e16, elo = (torch.zeros(Mh, Ca + Cb, device=DEV, dtype=dt) for dt in (torch.float16, torch.uint8) * 2)
            hip.concat_add(aa, Ca, ss, cc, Cb, Mh, None, h16, hlo)
            emu.concat_add(aa, Ca, ss, cc, Cb, Mh, None, e16, elo)
            src = torch.cat([aa, ss + cc], 1)
        else:
            h16, hlo, e16, elo = (torch.zeros(Mh, Ca, device=DEV, dtype=dt) for dt in (torch.float16, torch.uint8) * 2)
            other = rnd(Mh, Ca, seed=5) if fn == "add" else None
            if fn == "add":
                hip.add_f32(aa, other, Mh * Ca, None, h16, hlo)
                emu.add_f32(aa, other, Mh * Ca, None, e16, elo)
                src = aa + other
            else:
                hip.cast_f16(aa, Mh * Ca, h16, hlo)
                emu.cast_f16(aa, Mh * Ca, e16, elo)
                src = aa
        torch.cuda.synchronize()
        assert torch.equal(h16, e16) and torch.equal(h16, src.half())
        _same_bytes(fn, hlo, elo)
        big = src.abs() >= 512                      # lo clamps at 448 there; everywhere else the pair carries ~15 bits
        err = (rec(h16, hlo) - src).abs()
        assert (err[~big] <= 2.0 ** -14 * src.abs()[~big] + 2.0 ** -20).all()
    # norms: the fp32 value behind the byte is the kernel's own -> compare the reconstructed operand with the fp64 norm
    F, Npix, C, ppc = 2, 300, 320, 128
    x = rnd(F * Npix, C) * 1.7 + 0.9
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    nchunk = (Npix + ppc - 1) // ppc
    part = torch.zeros(F * nchunk * 32 * 3, device=DEV)
    y0, y1 = (torch.zeros(F * Npix, C, device=DEV, dtype=torch.float16) for _ in range(2))
    lo = torch.zeros(F * Npix, C, device=DEV, dtype=torch.uint8)
    hip.groupnorm_stats(x, C, F, Npix, C, ppc, part)
    hip.groupnorm_apply(x, C, F, Npix, C, ppc, part, gamma, beta, 1e-5, 1, y0, C)
    hip.groupnorm_apply(x, C, F, Npix, C, ppc, part, gamma, beta, 1e-5, 1, y1, C, lo)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.view(F, Npix, C).permute(0, 2, 1).double(), 32,
                                                                  gamma.double(), beta.double(), 1e-5)).permute(0, 2, 1).reshape(-1, C)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    check("gn hi+lo8", rec(y1, lo), ref.float(), 6e-5, 6e-5)
    assert (rec(y1, lo) - ref.float()).abs().max() < 0.2 * (y1.float() - ref.float()).abs().max()
    B, T = 1, 2
    xt = rnd(B * T * Npix, C) * 1.3 - 0.4
    t1, e1 = (torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.float16) for _ in range(2))
    tlo, elo = (torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.uint8) for _ in range(2))
    hip.groupnorm_temporal_silu(xt, B, T, Npix, C, gamma, beta, 1e-5, t1, tlo)
    emu.groupnorm_temporal_silu(xt, B, T, Npix, C, gamma, beta, 1e-5, e1, elo)
    torch.cuda.synchronize()
    check("gnt hi+lo8", rec(t1, tlo), rec(e1, elo), 6e-5, 6e-5)


@pytest.mark.parametrize("variant", ["o16", "res+o32+o16", "generic", "splitk"])
def test_gemm_writes_e4m3_lo_plane(variant):
    """out16_lo as e4m3 from the fast epilogue, the scalar (generic) epilogue and the split-K reduce"""
    if variant == "splitk":
        M, N, K = 256, 1280, 5120
        assert _splits(M=M, N=N, K=K) >= 2
    elif variant == "generic":
        M, N, K = 130, 100, 72
    else:
        M, N, K = 700, 640, 320
    a = rnd(M, K, dtype=torch.float16, seed=1)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
    res = rnd(M, N, seed=3)

    def outs():
        return dict(o32=torch.zeros(M, N, device=DEV), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16),
                    lo=torch.zeros(M, N, device=DEV, dtype=torch.uint8))

    def kw(o):
        k = dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, out16=o["o16"], ldc16=N, out16_lo=o["lo"])
        if variant != "o16":
            k.update(res1=res, ldr1=N, out32=o["o32"], ldc32=N)
        return k
    h, e = _run_both("gemm", outs, kw)
    check("out16", h["o16"], e["o16"], 4e-3)
    # the byte follows the kernel's own fp32 value: check the pair against the fp32 output where there is one
    rec = h["o16"].float() + _deq(h["lo"]) / 2048.0
    if variant != "o16":
        assert ((rec - h["o32"]).abs() <= 2.0 ** -14 * h["o32"].abs() + 2.0 ** -20).all()
    else:
        check("hi+lo8 vs emu pair", rec, e["o16"].float() + _deq(e["lo"]) / 2048.0, 3e-4, 0)


@pytest.mark.parametrize("tile", list(TILES))
@pytest.mark.parametrize("M,N,K,lda", [(1000, 1280, 320, 320), (700, 1280, 1344, 1360), (300, 320, 336, 336), (2048, 640, 2560, 2560)])
def test_gemm_e4m3_lo_pass_plain(M, N, K, lda, tile):
    """A = hi + e4m3(lo) * 2^-11 against W and its e4m3 copy: tracks the emulation to fp32 summation noise and the fp64
    product of the UNSPLIT activations far closer than the plain fp16 operand (K = 320: 2.5 lo tiles; 1344: a half tile of
    tail; 336: a 16-channel tail chunk; lda > K)."""
    a32 = torch.zeros(M, lda, device=DEV)
    a32[:, :K] = rnd(M, K, seed=7) * 3.0
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=8)
    hi, lo8 = _split8(a32)
    wl = engine.pk_lo8(w)
    ref = (a32[:, :K].double() @ w.double().t()).float()
    o_p, o_h, e_p = (torch.zeros(M, N, device=DEV) for _ in range(3))
    prev = hip.set_option(hip.OPT_GEMM_TILE, TILES[tile])
    try:
        hip.gemm(hi, w, M=M, N=N, K=K, lda=lda, out32=o_p, ldc32=N, a16_lo=lo8, w_lo=wl)
        hip.gemm(hi, w, M=M, N=N, K=K, lda=lda, out32=o_h, ldc32=N)
    finally:
        hip.set_option(hip.OPT_GEMM_TILE, prev)
    emu.gemm(hi, w, M=M, N=N, K=K, lda=lda, out32=e_p, ldc32=N, a16_lo=lo8, w_lo=wl)
    torch.cuda.synchronize()
    check("e4m3 lo pass vs emu", o_p, e_p, 2e-5, 1e-5)
    err_p, err_h = (o_p - ref).abs().max().item(), (o_h - ref).abs().max().item()
    print(f"e4m3-lo {err_p:.3e}  fp16 {err_h:.3e}")
    assert err_h > 1e-4 and err_p < err_h / 8


def test_gemm_e4m3_lo_pass_gathers_and_splitk():
    for Cin, stride, up in ((64, 1, 0), (128, 2, 0), (64, 1, 1)):
        F, H, W, N = 2, 10, 12, 320
        Ho, Wo = (2 * H, 2 * W) if up else ((H - 1) // stride + 1, (W - 1) // stride + 1)
        M, K = F * Ho * Wo, 9 * Cin
        x32 = rnd(F, H, W, Cin, seed=3) * 2.0
        hi, lo8 = _split8(x32)
        w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=4)
        wl = engine.pk_lo8(w)
        conv = dict(Cin=Cin, Hin=H, Win=W, Hout=Ho, Wout=Wo, stride=stride, upsample=up)
        prev = hip.set_option(hip.OPT_STENCIL_TILES, 2)      # the halo-tile kernel must decline an e4m3 plane
        try:
            h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
                a16=hi, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, out32=o["o"], ldc32=N, a16_lo=lo8, w_lo=wl))
        finally:
            hip.set_option(hip.OPT_STENCIL_TILES, prev)
        check(f"conv3x3 e4m3-lo Cin={Cin} s={stride} up={up}", h["o"], e["o"], 2e-5, 1e-5)
    for (B, T, Npix, C) in ((1, 8, 40, 1280), (2, 8, 300, 320), (1, 3, 77, 64)):   # K = 3840: split-K regime
        M, N, K = B * T * Npix, C, 3 * C
        x32 = rnd(M, C, seed=5)
        hi, lo8 = _split8(x32)
        w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=6)
        wl = engine.pk_lo8(w)
        res = rnd(M, N, seed=7)
        h, e = _run_both("gemm", lambda: dict(o=res.clone()), lambda o: dict(
            a16=hi, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=rnd(N, seed=8),
            res1=o["o"], ldr1=N, out32=o["o"], ldc32=N, a16_lo=lo8, w_lo=wl))
        check(f"conv1d_t e4m3-lo C={C}", h["o"], e["o"], 3e-5, 1e-5)
    # narrow output head (N = 4: 128x32 tile, scalar epilogue) and plain split-K
    F, H, W, Cin, N = 2, 16, 24, 320, 4
    M, K = F * H * W, 9 * Cin
    hi, lo8 = _split8(rnd(F, H, W, Cin, seed=9))
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=10)
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
        a16=hi, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, out32=o["o"], ldc32=N, a16_lo=lo8, w_lo=engine.pk_lo8(w)))
    check("head conv e4m3-lo", h["o"], e["o"], 2e-5, 1e-5)
    M, N, K = 256, 1280, 5120
    assert _splits(M=M, N=N, K=K) >= 2
    hi, lo8 = _split8(rnd(M, K, seed=11))
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=12)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
        a16=hi, w16=w, M=M, N=N, K=K, lda=K, out32=o["o"], ldc32=N, a16_lo=lo8, w_lo=engine.pk_lo8(w)))
    check("plain split-K e4m3-lo", h["o"], e["o"], 3e-5, 1e-5)


def test_e4m3_lo_pass_argument_checks():
    M, N, K = 128, 128, 72                       # K % 16 != 0
    a = rnd(M, K, dtype=torch.float16)
    w = rnd(N, K, dtype=torch.float16)
    lo8 = torch.zeros(M, K, device=DEV, dtype=torch.uint8)
    o = torch.zeros(M, N, device=DEV)
    with pytest.raises(hip.PncError):
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o, ldc32=N, a16_lo=lo8, w_lo=engine.pk_lo8(w))
    K = 64
    a, w, lo8 = a[:, :64].contiguous(), w[:, :64].contiguous(), lo8[:, :64].contiguous()
    with pytest.raises(hip.PncError):            # no e4m3 weights
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o, ldc32=N, a16_lo=lo8)
    with pytest.raises(hip.PncError):            # a lo plane is fp16 or uint8
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o, ldc32=N, a16_lo=lo8.float())
    hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o, ldc32=N, a16_lo=lo8, w_lo=engine.pk_lo8(w))
    torch.cuda.synchronize()
