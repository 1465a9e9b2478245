"""Per-kernel parity on the MI355X: every C-ABI entry point of libpanacea_hip.so against the plain
torch emulation (tests/emu.py, fp32 math on the same device tensors).

Inputs are asymmetric random data (a transposed fragment map cannot pass); shapes cover the tile
guards (ragged M/N/K), every A-gather mode, every epilogue flag, the view quirks and T < 8.
"""
import math

import pytest
import torch

import emu
from panacea_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, scale=1.0, dtype=torch.float32, seed=None):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xFFFF) + 17)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def check(name, got, ref, atol, rtol=2e-3):
    got, ref = got.float(), ref.float()
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    worst = (err - tol).max().item()
    if worst > 0:
        idx = torch.nonzero(err > tol)
        first = idx[0].tolist()
        raise AssertionError(
            f"{name}: {idx.shape[0]}/{err.numel()} elements off; max|err|={err.max().item():.4e} "
            f"(ref max {ref.abs().max().item():.3e}); first bad index {first}: got "
            f"{got[tuple(first)].item():.5f} ref {ref[tuple(first)].item():.5f}")


# ------------------------------------------------------------------------------------------- GEMM
def _run_both(fn_name, make_outs, kwargs_fn):
    outs_h, outs_e = make_outs(), make_outs()
    getattr(hip, fn_name)(**kwargs_fn(outs_h))
    getattr(emu, fn_name)(**kwargs_fn(outs_e))
    torch.cuda.synchronize()
    return outs_h, outs_e


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (128, 256, 128), (300, 320, 320), (130, 96, 72),
                                   (1000, 32, 64), (77, 16, 216), (16, 1280, 320), (3072, 640, 1920),
                                   (700, 960, 320), (513, 320, 1280)])
def test_gemm_plain_bias_out32_out16(M, N, K):
    a = rnd(M, K, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias = rnd(N)

    def outs():
        return dict(o32=torch.zeros(M, N, device=DEV), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))
    h, e = _run_both("gemm", outs, lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, out32=o["o32"], ldc32=N, out16=o["o16"], ldc16=N))
    check("out32", h["o32"], e["o32"], 2e-3)
    check("out16", h["o16"], e["o16"], 4e-3)


def test_gemm_identity_asymmetric():
    # A = I, asymmetric W: catches a transposed C fragment map or swapped operands exactly
    M = N = K = 128
    a = torch.eye(M, device=DEV, dtype=torch.float16)
    w = (torch.arange(N * K, device=DEV).view(N, K) % 251).to(torch.float16) / 16
    o = torch.zeros(M, N, device=DEV)
    hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o, ldc32=N)
    torch.cuda.synchronize()
    assert torch.equal(o, w.float().t().contiguous())


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (200, 128, 64)])
def test_gemm_epilogue_rowbias_residuals_inplace(M, N, K):
    a = rnd(M, K, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, rowb = rnd(N), rnd(4, N)
    res1_init, res2 = rnd(M, N + 8), rnd(M, N)

    def outs():
        return dict(o32=res1_init.clone(), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))
    # res1 aliases out32 (in-place residual stream), leading dimension N+8
    h, e = _run_both("gemm", outs, lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, rowbias=rowb, rb_rows=50, rb_mod=4,
        res1=o["o32"], ldr1=N + 8, res2=res2, ldr2=N, out32=o["o32"], ldc32=N + 8, out16=o["o16"], ldc16=N))
    check("out32", h["o32"], e["o32"], 2e-3)
    check("out16", h["o16"], e["o16"], 6e-3)


def test_gemm_act_silu():
    M, N, K = 256, 96, 216
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias = rnd(N)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV, dtype=torch.float16)), lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, act=hip.ACT_SILU, out16=o["o"], ldc16=N))
    check("silu", h["o"], e["o"], 4e-3)


@pytest.mark.parametrize("M,C", [(256, 64), (384, 320), (300, 128)])
def test_gemm_geglu(M, C):
    N, K = 8 * C, C
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias = rnd(N)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N // 2, device=DEV, dtype=torch.float16)), lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o["o"], ldc16=N // 2))
    check("geglu", h["o"], e["o"], 4e-3)


@pytest.mark.parametrize("M,C", [(512, 320), (1000, 640), (300, 128), (196608 // 8, 320)])
def test_gemm_geglu_register_epilogue_equals_the_staged_one(M, C):
    """Round 3: without a lo plane the GEGLU epilogue does the gate arithmetic on the accumulators and stages only the fp16 result;
    with one (out16_lo) it stages the fp32 value / gate blocks as before.  Same products, same roundings (the register path keeps
    its last product opaque so that hipcc does not fuse multiply + conversion into one rounding — measured: that alone moved 42 of
    655 360 elements by one fp16 ulp): the hi planes are equal bit for bit, ragged row tails included."""
    N, K = 8 * C, C
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias = rnd(N)
    o_reg = torch.zeros(M, N // 2, device=DEV, dtype=torch.float16)
    o_stg, lo = torch.zeros_like(o_reg), torch.zeros_like(o_reg)
    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o_reg, ldc16=N // 2)
    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o_stg, ldc16=N // 2, out16_lo=lo)
    torch.cuda.synchronize()
    assert torch.equal(o_reg, o_stg)
    assert o_reg.abs().max().item() > 0.1 and lo.abs().max().item() > 0
    o_nb = torch.zeros_like(o_reg)
    hip.gemm(a, w, M=M, N=N, K=K, lda=K, geglu=True, out16=o_nb, ldc16=N // 2)          # no bias
    e_nb = torch.zeros_like(o_reg)
    emu.gemm(a, w, M=M, N=N, K=K, lda=K, geglu=True, out16=e_nb, ldc16=N // 2)
    check("geglu no bias", o_nb, e_nb, 4e-3)


@pytest.mark.parametrize("M,C,bias", [(256 * 64, 320, True), (256 * 26, 640, True), (256 * 70, 320, False)])
def test_gemm_geglu_persistent_kernel_is_bit_identical(M, C, bias):
    """PNC_OPT_GEMM_PERSIST (the default): one workgroup per CU walks its output tiles and requests the next tile's first K tile before the
    current tile's epilogue.  Same tiles, same K order, same epilogue arithmetic: bit-identical to the one-tile-per-workgroup launch
    (shapes with >= 512 full 256x256 tiles qualify; others keep the regular kernel whatever the switch says)."""
    N, K = 8 * C, C
    assert (M // 256) * (N // 256) >= 512
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    b = rnd(N) if bias else None
    o_ref = torch.zeros(M, N // 2, device=DEV, dtype=torch.float16)
    o_per = torch.zeros_like(o_ref)
    prev = hip.set_option(hip.OPT_GEMM_PERSIST, 0)
    try:
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=b, geglu=True, out16=o_ref, ldc16=N // 2)
        hip.set_option(hip.OPT_GEMM_PERSIST, 1)
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=b, geglu=True, out16=o_per, ldc16=N // 2)
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=b, geglu=True, out16=o_per, ldc16=N // 2)      # twice: no state left behind
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_GEMM_PERSIST, prev)
    assert torch.equal(o_per, o_ref) and o_ref.abs().max().item() > 0.1


@pytest.mark.parametrize("variant", ["res_ln", "res_ln_lo8", "ln_lo8", "pos_ln", "res", "res_lo8", "o32", "ff2_o16_lo8", "q_o16", "qkv_vt"])
def test_gemm_persistent_kernel_is_bit_identical(variant):
    """PNC_OPT_GEMM_PERSIST bit 1 (round 4): plain-A launches of >= 512 full 256x320 tiles run as one persistent workgroup per CU
    that requests the next output tile's first K tile before its epilogue; the epilogue's LDS staging is inline asm (it shares the
    operand ring with the DMA in flight).  Same tiles, K order, MFMA order, epilogue arithmetic (incl. the fused LayerNorm's
    one-pass statistics): bit-identical to the one-tile-per-workgroup kernel for every epilogue variant the network uses at
    level 0 — residual in place + LayerNorm, proj_in (+ position table) + norm1 with an e4m3 lo pass, FF2 with fp16 + e4m3
    outputs, the q projection, QKV with the channel-major V^T."""
    from panacea_amd import engine
    C = 320
    K = 4 * C if variant.startswith("ff2") else C
    N = 3 * C if variant == "qkv_vt" else C
    M = 256 * (512 if N == C else 176)
    assert (M // 256) * (N // 320) >= 512
    a32 = rnd(M, K, seed=41)
    a = a32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=42)
    bias = rnd(N, seed=43)
    kw = dict(M=M, N=N, K=K, lda=K, bias=bias)
    if "lo8" in variant:
        alo = torch.zeros(M, K, device=DEV, dtype=torch.uint8)
        hip.cast_f16(a32, M * K, torch.zeros_like(a), alo)
        kw.update(a16_lo=alo, w_lo=engine.pk_lo8(w))
    gamma, beta = rnd(N, seed=44) * 0.5 + 1, rnd(N, seed=45) * 0.3
    res0 = rnd(M, N, seed=46)

    def run():
        o = {}
        k2 = dict(kw)
        if variant in ("res_ln", "res_ln_lo8", "res", "res_lo8"):
            o["x"] = res0.clone()                                  # the residual stream, updated in place
            k2.update(res1=o["x"], ldr1=N, out32=o["x"], ldc32=N)
        if variant in ("ln_lo8", "pos_ln", "o32"):
            o["x"] = torch.zeros(M, N, device=DEV)
            k2.update(out32=o["x"], ldc32=N)
        if variant == "pos_ln":
            k2.update(rowbias=rnd(8, N, seed=47), rb_rows=M // 16, rb_mod=8)
        if "ln" in variant:
            o["ln"] = torch.zeros(M, N, device=DEV, dtype=torch.float16)
            k2.update(ln_gamma=gamma, ln_beta=beta, ln_out16=o["ln"], ldln=N)
        if variant == "ff2_o16_lo8":
            o["h"], o["lo"] = torch.zeros(M, N, device=DEV, dtype=torch.float16), torch.zeros(M, N, device=DEV, dtype=torch.uint8)
            k2.update(res1=res0, ldr1=N, out16=o["h"], ldc16=N, out16_lo=o["lo"])
        if variant == "q_o16":
            o["h"] = torch.zeros(M, N, device=DEV, dtype=torch.float16)
            k2.update(out16=o["h"], ldc16=N)
        if variant == "qkv_vt":
            t_rows = M // 4
            o["qk"], o["vt"] = torch.zeros(M, 2 * C, device=DEV, dtype=torch.float16), torch.zeros(4, C, t_rows, device=DEV, dtype=torch.float16)
            k2.update(out16=o["qk"], ldc16=2 * C, out16t=o["vt"], ldt=t_rows, t_rows=t_rows, t_gstride=C * t_rows, n_split=2 * C)
            k2["bias"] = None
        hip.gemm(a, w, **k2)
        return o
    prev = hip.set_option(hip.OPT_GEMM_PERSIST, 1)
    try:
        ref = run()
        hip.set_option(hip.OPT_GEMM_PERSIST, 3)
        got = run()
        got2 = run()                                               # twice: no state left behind
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_GEMM_PERSIST, prev)
    for k in ref:
        assert torch.equal(got[k], ref[k]) and torch.equal(got2[k], ref[k]), (variant, k, (got[k].float() - ref[k].float()).abs().max().item())
        assert ref[k].float().abs().max().item() > 0.1


@pytest.mark.parametrize("kind,M,N,K", [
    ("plain", 1024, 640, 192), ("plain", 768, 1280, 128), ("plain", 1000, 640, 128),      # 256x320 / 256x256 tiles; a ragged last row tile
    ("plain_lo8", 1024, 320, 320), ("persist", 256 * 512, 320, 128), ("persist_lo8", 256 * 512, 320, 320),
    ("conv3x3", 2 * 16 * 32, 320, 9 * 64), ("stencil", 2 * 16 * 32, 320, 9 * 64),
])
@pytest.mark.parametrize("res", [False, True])
def test_gemm_direct_fp32_epilogue_is_bit_identical(kind, M, N, K, res):
    """Round 6: fp32-only epilogues (out32 = acc + bias [+ res1, in place]) of full 256-row tiles leave the accumulators through buffer
    stores (and the residual comes in through buffer loads) instead of through the LDS staging — epi_direct_o32 in gemm_glds_kernel,
    gemm_persist_kernel and stencil_tile_kernel.  Same additions in the same order: bit-identical to the staged epilogue
    (PNC_OPT_GEMM_FUSE_LN + 2), incl. the tiles that fall back to it (ragged rows) and the e4m3 lo pass in front."""
    from panacea_amd import engine
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=72)
    bias = rnd(N, seed=73)
    res0 = rnd(M, N, seed=74)
    kw = dict(M=M, N=N, K=K, bias=bias)
    if kind in ("conv3x3", "stencil"):
        Cin = K // 9
        x = rnd(2, 16, 32, Cin, dtype=torch.float16, seed=71)
        kw.update(a16=x, a_mode=hip.A_CONV3X3, conv=dict(Cin=Cin, Hin=16, Win=32, Hout=16, Wout=32, stride=1, upsample=0))
    else:
        a32 = rnd(M, K, seed=71)
        kw.update(a16=a32.half(), lda=K)
        if "lo8" in kind:
            alo = torch.zeros(M, K, device=DEV, dtype=torch.uint8)
            hip.cast_f16(a32, M * K, torch.zeros(M, K, device=DEV, dtype=torch.float16), alo)
            kw.update(a16_lo=alo, w_lo=engine.pk_lo8(w))

    def run(fuse_opt):
        o = res0.clone() if res else torch.zeros(M, N, device=DEV)
        k2 = dict(kw, w16=w, out32=o, ldc32=N)
        if res:
            k2.update(res1=o, ldr1=N)
        prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, fuse_opt)
        pst = hip.set_option(hip.OPT_STENCIL_TILES, 2 if kind == "stencil" else 0)
        ppe = hip.set_option(hip.OPT_GEMM_PERSIST, 3 if kind.startswith("persist") else 1)
        try:
            hip.gemm(**k2)
            torch.cuda.synchronize()
        finally:
            hip.set_option(hip.OPT_GEMM_FUSE_LN, prev)
            hip.set_option(hip.OPT_STENCIL_TILES, pst)
            hip.set_option(hip.OPT_GEMM_PERSIST, ppe)
        return o
    staged, direct, again = run(3), run(1), run(1)
    assert torch.equal(direct, staged), (kind, res, (direct - staged).abs().max().item())
    assert torch.equal(again, staged)
    assert staged.abs().max().item() > 0.1


@pytest.mark.parametrize("G,t_rows,C", [(2, 128, 64), (3, 96, 128), (2, 80, 64), (2, 192, 320)])
def test_gemm_split_transposed_output(G, t_rows, C):
    # q|k row-major for n < 2C, V^T channel-major for n >= 2C  (QKV projection epilogue)
    M, N, K = G * t_rows, 3 * C, C
    ldt = t_rows + 8
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)

    def outs():
        return dict(qk=torch.zeros(M, 2 * C, device=DEV, dtype=torch.float16),
                    vt=torch.zeros(G, C, ldt, device=DEV, dtype=torch.float16))
    h, e = _run_both("gemm", outs, lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, out16=o["qk"], ldc16=2 * C, out16t=o["vt"], ldt=ldt,
        t_rows=t_rows, t_gstride=C * ldt, n_split=2 * C))
    check("qk", h["qk"], e["qk"], 4e-3)
    check("vt", h["vt"], e["vt"], 4e-3)


@pytest.mark.parametrize("F,Hin,Win,Cin,N,stride,up", [
    (2, 8, 24, 8, 64, 1, 0), (2, 8, 24, 64, 64, 2, 0), (1, 4, 12, 128, 128, 1, 1),
    (2, 16, 48, 24, 16, 1, 0), (1, 16, 48, 16, 32, 2, 0), (2, 9, 11, 32, 320, 1, 0), (1, 8, 8, 320, 8, 1, 0)])
def test_gemm_conv3x3(F, Hin, Win, Cin, N, stride, up):
    if up:
        Hout, Wout = 2 * Hin, 2 * Win
    else:
        Hout, Wout = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    M, K = F * Hout * Wout, 9 * Cin
    x = rnd(F, Hin, Win, Cin, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias = rnd(N)
    conv = dict(Cin=Cin, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride, upsample=up)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
        a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias, out32=o["o"], ldc32=N))
    check("conv3x3", h["o"], e["o"], 2e-3)


@pytest.mark.parametrize("B,T,Npix,C", [(2, 8, 48, 64), (1, 2, 100, 128), (2, 1, 64, 64), (1, 8, 32, 320)])
def test_gemm_conv1d_temporal(B, T, Npix, C):
    M, N, K = B * T * Npix, C, 3 * C
    x = rnd(M, C, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, emb = rnd(N), rnd(B * T, N)
    res = rnd(M, N)

    def outs():
        return dict(o=res.clone())
    h, e = _run_both("gemm", outs, lambda o: dict(
        a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=bias,
        rowbias=emb, rb_rows=Npix, rb_mod=B * T, res1=o["o"], ldr1=N, out32=o["o"], ldc32=N))
    check("conv1d_t", h["o"], e["o"], 2e-3)


@pytest.mark.parametrize("F,H,W,V,Cin,N,stride,up,lo", [
    (2, 16, 64, 2, 64, 320, 1, 0, None),        # stencil tiles (16 x 16)
    (1, 32, 192, 3, 128, 320, 1, 0, "f16"),     # stencil tiles, two slices, precise operand (fp16 lo plane)
    (3, 8, 96, 3, 64, 256, 1, 0, "e4m3"),       # 8 x 32 tiles / per-tap lo8 pass
    (2, 8, 48, 2, 16, 64, 1, 0, None),          # narrow input: the per-tap gather, K order (ky, kx, ci)
    (2, 16, 48, 2, 64, 128, 2, 0, None),        # stride 2 (Downsample): only the left column is read
    (1, 8, 48, 6, 128, 64, 1, 1, None),         # nearest x2 then conv (Upsample), six bands
    (2, 8, 96, 2, 320, 320, 2, 0, "e4m3"),      # Downsample of a precise stream
    (1, 16, 96, 2, 320, 640, 1, 1, "f16"),
])
def test_gemm_conv3x3_view_band_columns(F, H, W, V, Cin, N, stride, up, lo):
    """PncGemmParams.x_halo_off (round 4, engine.ViewShard): a 3x3 conv over a BAND of the panorama whose columns -1 / W_l sit in a
    block behind the band.  The bands' outputs, side by side, are the conv over the whole panorama — the same products in the same
    K order per output pixel (compared to 2e-6 of the output scale: only a split-K choice could reorder them) — and each band
    agrees with the emulation."""
    from panacea_amd import engine
    wl = W // V
    K = 9 * Cin
    x32 = rnd(F, H, W, Cin, seed=71)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=72)
    bias = rnd(N, seed=73)

    def planes(v32):
        hi = v32.half()
        if lo == "f16":
            return hi, ((v32 - hi.float()) * 2048.0).half()
        if lo == "e4m3":
            q = torch.zeros(v32.shape, device=DEV, dtype=torch.uint8)
            hip.cast_f16(v32.contiguous(), v32.numel(), torch.zeros_like(hi), q)
            return hi, q
        return hi, None
    w_lo = engine.pk_lo8(w) if lo == "e4m3" else None
    Hout, Wfull = (2 * H, 2 * W) if up else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    hi, lop = planes(x32)
    full = torch.zeros(F * Hout * Wfull, N, device=DEV)
    hip.gemm(a16=hi, w16=w, M=full.shape[0], N=N, K=K, a_mode=hip.A_CONV3X3, bias=bias, out32=full, ldc32=N, a16_lo=lop, w_lo=w_lo,
             conv=dict(Cin=Cin, Hin=H, Win=W, Hout=Hout, Wout=Wfull, stride=stride, upsample=up))
    Wout = Wfull // V
    zero = torch.zeros(F, H, 1, Cin, device=DEV)
    for v in range(V):
        band = x32[:, :, v * wl:(v + 1) * wl]
        left = x32[:, :, v * wl - 1:v * wl] if v > 0 else zero
        right = x32[:, :, (v + 1) * wl:(v + 1) * wl + 1] if v < V - 1 else zero
        whole = torch.cat([band.reshape(-1, Cin), left.reshape(-1, Cin), right.reshape(-1, Cin)])
        bhi, blo = planes(whole)
        kw = dict(a16=bhi, w16=w, M=F * Hout * Wout, N=N, K=K, a_mode=hip.A_CONV3X3, bias=bias, ldc32=N, a16_lo=blo, w_lo=w_lo,
                  conv=dict(Cin=Cin, Hin=H, Win=wl, Hout=Hout, Wout=Wout, stride=stride, upsample=up, x_halo_off=F * H * wl * Cin))
        oh, oe = torch.zeros(kw["M"], N, device=DEV), torch.zeros(kw["M"], N, device=DEV)
        hip.gemm(out32=oh, **kw)
        emu.gemm(out32=oe, **kw)
        torch.cuda.synchronize()
        check(f"band {v} vs emu", oh, oe, *((2e-5, 1e-5) if lo else (2e-3, 2e-3)))
        want = full.view(F, Hout, Wfull, N)[:, :, v * Wout:(v + 1) * Wout].reshape(-1, N)
        check(f"band {v} vs the panorama's conv", oh, want, 2e-6, 2e-6)


@pytest.mark.parametrize("B,Tl,Npix,C,lo8", [(2, 2, 96, 64, False), (1, 4, 300, 320, True), (2, 1, 64, 128, False), (1, 2, 3072, 320, True)])
def test_gemm_conv1d_temporal_halo_layout(B, Tl, Npix, C, lo8):
    """PncGemmParams.t_halo (round 4, engine.FrameShard): A holds Tl + 2 frames per sample — the frame before and after the Tl the
    rows speak of are a neighbour rank's (or zeros) — and no tap is padded.  Against the emulation, and BIT-equal to the plain
    launch over the Tl + 2 frames restricted to the inner frames (same products in the same order)."""
    from panacea_amd import engine
    M, N, K = B * Tl * Npix, C, 3 * C
    x32 = rnd(B * (Tl + 2) * Npix, C, seed=61)
    x = x32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=62)
    bias, emb, res = rnd(N, seed=63), rnd(B * Tl, N, seed=64), rnd(M, N, seed=65)
    kw = dict(a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=Tl, Npix=Npix, halo=1), bias=bias,
              rowbias=emb, rb_rows=Npix, rb_mod=B * Tl, ldr1=N, ldc32=N)
    xlo = None
    if lo8:
        xlo = torch.zeros(B * (Tl + 2) * Npix, C, device=DEV, dtype=torch.uint8)
        hip.cast_f16(x32, x32.numel(), torch.zeros_like(x), xlo)
        kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))
    oh, oe = res.clone(), res.clone()
    hip.gemm(res1=oh, out32=oh, **kw)
    emu.gemm(res1=oe, out32=oe, **kw)
    torch.cuda.synchronize()
    check("conv1d_t_halo", oh, oe, 2e-3)
    # the plain launch over all Tl + 2 frames computes the same inner frames (its outer two see zero padding instead of halos)
    Mf = B * (Tl + 2) * Npix
    full = torch.zeros(Mf, N, device=DEV)
    kf = dict(kw, M=Mf, tconv=dict(C=C, T=Tl + 2, Npix=Npix), rowbias=None, rb_rows=0, rb_mod=0)
    hip.gemm(out32=full, **kf)
    plain = torch.zeros(M, N, device=DEV)
    hip.gemm(out32=plain, **dict(kw, rowbias=None, rb_rows=0, rb_mod=0))
    torch.cuda.synchronize()
    assert torch.equal(plain.view(B, Tl, Npix, N), full.view(B, Tl + 2, Npix, N)[:, 1:Tl + 1])


@pytest.mark.parametrize("B,T,Npix,C,epi,lo8", [
    (2, 8, 12288, 320, "rb", True),        # level 0: 10 channels per group, chunks of 8 columns straddle the groups
    (2, 8, 3072, 640, "r2", True),         # level 1: 20 per group, 384 tiles = a full round + tail-split tiles (quarter workgroups)
    (2, 8, 768, 1280, "r2+o16", False),    # level 2: 40 per group (256-column tiles are chosen here: the statistics launch follows)
    (2, 8, 192, 1280, "rb", True),         # level 3: 192 pixels per frame, small tiles: the statistics launch follows
    (1, 4, 1024, 320, "r2+o16", False),
    (1, 2, 640, 640, "rb", False),         # 640 pixels = 2.5 tiles per frame: a tile's wave blocks belong to two frames
])
def test_gemm_conv1d_groupnorm_records_from_the_epilogue(B, T, Npix, C, epi, lo8):
    """PncGemmParams.gn_part (round 4): the temporal conv writes the GroupNorm(32) records of its fp32 output — one {n, mean, M2}
    per (64-pixel block, group) — out of its epilogue.  The output is unchanged to the bit; the records combine to the statistics
    pnc_groupnorm_stats computes from the output (and feed pnc_groupnorm_apply through n_records); they are bit-reproducible; and
    with PNC_OPT_GEMM_GN_STATS = 0 the library's trailing statistics launch fills the same buffer."""
    from panacea_amd import engine
    M, N, K = B * T * Npix, C, 3 * C
    x32 = rnd(M, C, seed=81)
    x = x32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=82)
    bias, emb = rnd(N, seed=83), rnd(B * T, N, seed=84)
    res, skip = rnd(M, N, seed=85) * 2.0 + 0.5, rnd(M, N, seed=86)
    kw = dict(a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=bias, ldr1=N, ldc32=N)
    if lo8:
        xlo = torch.zeros(M, C, device=DEV, dtype=torch.uint8)
        hip.cast_f16(x32, x32.numel(), torch.zeros_like(x), xlo)
        kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))
    if epi == "rb":
        kw.update(rowbias=emb, rb_rows=Npix, rb_mod=B * T)
    else:
        kw.update(res2=skip, ldr2=N)
    o16 = torch.zeros(M, N, device=DEV, dtype=torch.float16) if "o16" in epi else None
    if o16 is not None:
        kw.update(out16=o16, ldc16=N)
    F, nrec = B * T, -(-Npix // 64)

    def run(opt, with_part):
        prev = hip.set_option(hip.OPT_GEMM_GN_STATS, opt)
        try:
            out = res.clone()
            part = torch.full((F * nrec * 96,), float("nan"), device=DEV) if with_part else None
            hip.gemm(res1=out, out32=out, gn_part=part, **kw)
            torch.cuda.synchronize()
        finally:
            hip.set_option(hip.OPT_GEMM_GN_STATS, prev)
        return out, part
    plain, _ = run(1, False)
    fused, part = run(1, True)
    again, part2 = run(1, True)
    trail, part_t = run(0, True)
    assert torch.equal(fused, plain) and torch.equal(trail, plain)                 # the output does not know about the records
    assert torch.equal(part, part2)                                               # fixed summation order
    assert torch.isfinite(part).all() and torch.isfinite(part_t).all()
    # the records against float64 statistics of the output, per (frame, 64-pixel block, group)
    blk = plain.double().view(F, Npix, 32, C // 32)
    P = part.view(F, nrec, 32, 3).double()
    for r in sorted({0, 1, 2, nrec // 2, nrec - 2, nrec - 1} & set(range(nrec))):
        v = blk[:, r * 64:(r + 1) * 64]
        n = v.shape[1] * (C // 32)
        assert torch.equal(P[:, r, :, 0], torch.full((F, 32), float(n), device=DEV, dtype=torch.float64))
        mean, m2 = v.mean(dim=(1, 3)), v.var(dim=(1, 3), unbiased=False) * n
        assert (P[:, r, :, 1] - mean).abs().max().item() <= 2e-6 * (1 + mean.abs().max().item()), r
        assert ((P[:, r, :, 2] - m2).abs() / m2).max().item() <= 2e-4, r
    Pt = part_t.view(F, nrec, 32, 3).double()
    assert (P[..., 1] - Pt[..., 1]).abs().max().item() <= 4e-6 * (1 + Pt[..., 1].abs().max().item())
    assert ((P[..., 2] - Pt[..., 2]).abs() / Pt[..., 2]).max().item() <= 4e-4
    # ... and through the apply kernel: same normalised operand as the two-launch GroupNorm of the output (fp16 roundings may flip)
    gamma, beta = rnd(C, seed=87) * 0.2 + 1.0, rnd(C, seed=88) * 0.1
    y_ep, y_ref = torch.zeros(M, C, device=DEV, dtype=torch.float16), torch.zeros(M, C, device=DEV, dtype=torch.float16)
    ppc = engine._ppc(Npix)
    hip.groupnorm_apply(plain, C, F, Npix, C, ppc, part, gamma, beta, 1e-5, True, y_ep, C, n_records=nrec)
    pr = torch.zeros(F * (-(-Npix // ppc)) * 96, device=DEV)
    hip.groupnorm_stats(plain, C, F, Npix, C, ppc, pr)
    hip.groupnorm_apply(plain, C, F, Npix, C, ppc, pr, gamma, beta, 1e-5, True, y_ref, C)
    torch.cuda.synchronize()
    check("GroupNorm from the epilogue's records", y_ep, y_ref, 2e-3, 2e-3)
    assert (y_ep != y_ref).float().mean().item() < 2e-2


@pytest.mark.parametrize("B,T,Tl,Npix,C,lo8", [(2, 8, 2, 96, 64, False), (1, 4, 4, 77, 320, True), (2, 8, 4, 64, 1280, True), (1, 2, 1, 40, 128, False)])
def test_groupnorm_temporal_in_parts(B, T, Tl, Npix, C, lo8):
    """pnc_groupnorm_temporal_part (round 4): the T frames of a pixel on T / Tl ranks — every rank's partial {sum, sum of squares}
    (mode 1), added up, then the normalisation of the local frames (mode 2) into the (Tl + 2)-frame layout — against the fused
    kernel over all T frames"""
    x = rnd(B * T * Npix, C, seed=71) * 1.3 - 0.4
    gamma, beta = rnd(C, seed=72) * 0.5 + 1, rnd(C, seed=73) * 0.3
    ref = torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.float16)
    ref_lo = torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.uint8) if lo8 else None
    hip.groupnorm_temporal_silu(x, B, T, Npix, C, gamma, beta, 1e-5, ref, ref_lo)
    G = T // Tl
    xv = x.view(B, T, Npix, C)
    parts = [xv[:, g * Tl:(g + 1) * Tl].contiguous() for g in range(G)]
    stats = [torch.zeros(B * Npix * 64, device=DEV) for _ in range(G)]
    for g in range(G):
        hip.groupnorm_temporal_part(parts[g], B, Tl, Npix, C, gamma, beta, 1e-5, stats[g], 1, T)
    total = torch.stack(stats).sum(0)
    es = torch.zeros_like(total)
    emu.groupnorm_temporal_part(parts[0], B, Tl, Npix, C, gamma, beta, 1e-5, es, 1, T)
    torch.cuda.synchronize()
    assert torch.allclose(stats[0], es, rtol=1e-5, atol=1e-4)
    for g in range(G):
        y = torch.full((B, Tl + 2, Npix, C), 9.0, device=DEV, dtype=torch.float16)
        ylo = torch.zeros(B, Tl + 2, Npix, C, device=DEV, dtype=torch.uint8) if lo8 else None
        hip.groupnorm_temporal_part(parts[g], B, Tl, Npix, C, gamma, beta, 1e-5, total, 2, T, y, ylo, 1)
        torch.cuda.synchronize()
        assert (y[:, 0] == 9.0).all() and (y[:, -1] == 9.0).all()          # the halo slots are the exchange's, not the kernel's
        want = ref.view(B, T, Npix, C)[:, g * Tl:(g + 1) * Tl]
        check(f"gn_temporal_part[{g}]", y[:, 1:Tl + 1], want, 2e-3)
        if lo8:
            rec = y[:, 1:Tl + 1].float() + ylo[:, 1:Tl + 1].view(torch.float8_e4m3fn).float() / 2048.0
            rr = want.float() + ref_lo.view(B, T, Npix, C)[:, g * Tl:(g + 1) * Tl].view(torch.float8_e4m3fn).float() / 2048.0
            check(f"gn_temporal_part_lo[{g}]", rec, rr, 2e-4, 1e-4)


@pytest.mark.parametrize("M,N,K,geglu", [(1000, 320, 640, False), (49152, 640, 640, False), (700, 1280, 320, True),
                                         (12288, 1280, 192, False)])
def test_gemm_tail_row_split_is_bit_identical(M, N, K, geglu):
    # the partial last round of tiles is run as half / quarter tiles (BM/2, BM/4 rows per workgroup): same bits
    a = rnd(M, K, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, res = rnd(N), rnd(M, N)
    No = N // 2 if geglu else N

    def run():
        o32 = res.clone() if not geglu else None
        o16 = torch.zeros(M, No, device=DEV, dtype=torch.float16)
        kw = dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, out16=o16, ldc16=No, geglu=geglu)
        if not geglu:
            kw.update(res1=o32, ldr1=N, out32=o32, ldc32=N)
        hip.gemm(**kw)
        torch.cuda.synchronize()
        return o32, o16
    s32, s16 = run()
    prev = hip.set_option(hip.OPT_GEMM_TAIL_SPLIT, 0)
    try:
        f32, f16 = run()
    finally:
        hip.set_option(hip.OPT_GEMM_TAIL_SPLIT, prev)
    assert torch.equal(s16, f16) and (geglu or torch.equal(s32, f32))
    e16 = torch.zeros(M, No, device=DEV, dtype=torch.float16)
    kw = dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, out16=e16, ldc16=No, geglu=geglu)
    if not geglu:
        e32 = res.clone()
        kw.update(res1=e32, ldr1=N, out32=e32, ldc32=N)
    emu.gemm(**kw)
    check("tail split vs emu", s16, e16, 6e-3)



# ---------------------------------------------------------- specialised epilogue variants x tile geometries
TILES = {"128x128": 1, "256x128": 2, "256x320": 3, "256x256": 4}


def _epi_case(name, M, N, K, seed=0):
    """kwargs factory of one fast-epilogue variant (16-byte aligned everything: the vector contract)"""
    a = rnd(M, K, dtype=torch.float16, seed=seed + 1)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=seed + 2)
    bias, res, res2, rowb = rnd(N, seed=seed + 3), rnd(M, N, seed=seed + 4), rnd(M, N, seed=seed + 5), rnd(4, N, seed=seed + 6)
    base = dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias)

    def outs():
        return dict(o32=res.clone(), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16),
                    o16lo=torch.zeros(M, N, device=DEV, dtype=torch.float16))
    kws = {
        "o16": lambda o: dict(base, out16=o["o16"], ldc16=N),
        "o16+lo": lambda o: dict(base, out16=o["o16"], ldc16=N, out16_lo=o["o16lo"]),
        "o32": lambda o: dict(base, out32=o["o32"], ldc32=N),
        "o32+o16": lambda o: dict(base, out32=o["o32"], ldc32=N, out16=o["o16"], ldc16=N),
        "r1+o32": lambda o: dict(base, res1=o["o32"], ldr1=N, out32=o["o32"], ldc32=N),
        "r1+o32+o16+lo": lambda o: dict(base, res1=o["o32"], ldr1=N, out32=o["o32"], ldc32=N, out16=o["o16"], ldc16=N,
                                        out16_lo=o["o16lo"]),
        "r1+o16": lambda o: dict(base, res1=res2, ldr1=N, out16=o["o16"], ldc16=N),
        "rb+o32": lambda o: dict(base, rowbias=rowb, rb_rows=50, rb_mod=4, out32=o["o32"], ldc32=N),
        "r2only+o32": lambda o: dict(base, res2=res2, ldr2=N, out32=o["o32"], ldc32=N),
    }
    return outs, kws[name]


@pytest.mark.parametrize("tile", list(TILES))
@pytest.mark.parametrize("name", ["o16", "o16+lo", "o32", "o32+o16", "r1+o32", "r1+o32+o16+lo", "r1+o16", "rb+o32",
                                  "r2only+o32"])
def test_gemm_fast_epilogue_variants(name, tile):
    # N = 1280 admits every geometry (1280 = 4 x 320 = 5 x 256); ragged M exercises the row predicate
    M, N, K = 1000, 1280, 128
    outs, kw = _epi_case(name, M, N, K)
    prev = hip.set_option(hip.OPT_GEMM_TILE, TILES[tile])
    try:
        h, e = _run_both("gemm", outs, kw)
    finally:
        hip.set_option(hip.OPT_GEMM_TILE, prev)
    check("out32", h["o32"], e["o32"], 2e-3)
    check("out16", h["o16"], e["o16"], 4e-3)
    # the lo plane carries (v - hi) * 2^11: compare the RECONSTRUCTED value with the fp32 output where both exist
    if "lo" in name:
        rec = h["o16"].float() + h["o16lo"].float() / 2048.0
        ref = e["o16"].float() + e["o16lo"].float() / 2048.0
        check("hi+lo", rec, ref, 2e-3, 2e-3)
        if "o32" in name:
            assert (rec - h["o32"]).abs().max().item() <= 2.0 ** -20 * max(1.0, h["o32"].abs().max().item())


@pytest.mark.parametrize("tile", list(TILES))
@pytest.mark.parametrize("mode", ["conv1d+rb", "conv1d+r2", "conv1d+r2+o16", "conv3x3+o32", "conv3x3+silu16"])
def test_gemm_fast_epilogue_variants_conv(mode, tile):
    prev = hip.set_option(hip.OPT_GEMM_TILE, TILES[tile])
    try:
        if mode.startswith("conv1d"):
            B, T, Npix, C = 1, 4, 150, 1280
            M, N, K = B * T * Npix, C, 3 * C
            x = rnd(M, C, dtype=torch.float16, seed=1)
            w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
            bias, emb, res, res2 = rnd(N, seed=3), rnd(B * T, N, seed=4), rnd(M, N, seed=5), rnd(M, N, seed=6)
            base = dict(a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=bias)

            def outs():
                return dict(o32=res.clone(), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))
            if mode == "conv1d+rb":
                kw = lambda o: dict(base, rowbias=emb, rb_rows=Npix, rb_mod=B * T, res1=o["o32"], ldr1=N, out32=o["o32"], ldc32=N)
            elif mode == "conv1d+r2":
                kw = lambda o: dict(base, res1=o["o32"], ldr1=N, res2=res2, ldr2=N, out32=o["o32"], ldc32=N)
            else:
                kw = lambda o: dict(base, res1=o["o32"], ldr1=N, res2=res2, ldr2=N, out32=o["o32"], ldc32=N,
                                    out16=o["o16"], ldc16=N)
        else:
            F, H, W, Cin, N = 2, 10, 24, 64, 1280
            M, K = F * H * W, 9 * Cin
            x = rnd(F, H, W, Cin, dtype=torch.float16, seed=1)
            w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
            conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
            base = dict(a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=rnd(N, seed=3))

            def outs():
                return dict(o32=torch.zeros(M, N, device=DEV), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))
            if mode == "conv3x3+o32":
                kw = lambda o: dict(base, out32=o["o32"], ldc32=N)
            else:
                kw = lambda o: dict(base, act=hip.ACT_SILU, out16=o["o16"], ldc16=N)
        h, e = _run_both("gemm", outs, kw)
    finally:
        hip.set_option(hip.OPT_GEMM_TILE, prev)
    check("out32", h["o32"], e["o32"], 2e-3)
    check("out16", h["o16"], e["o16"], 4e-3)


def _split(v):
    hi = v.half()
    return hi, ((v - hi.float()) * 2048.0).half()


@pytest.mark.parametrize("tile", list(TILES))
@pytest.mark.parametrize("M,N,K", [(1000, 1280, 320), (700, 1280, 1344)])
def test_gemm_precise_operand_plain(M, N, K, tile):
    """A = hi + lo * 2^-11: the result must track the fp64 product of the UNSPLIT fp32 activations ~2^-11 closer than
    the plain fp16 operand does (what the precise mode is for), on every tile geometry."""
    a32 = rnd(M, K, seed=7) * 3.0
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=8)
    hi, lo = _split(a32)
    ref = (a32.double() @ w.double().t()).float()
    o_p, o_h = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    prev = hip.set_option(hip.OPT_GEMM_TILE, TILES[tile])
    try:
        hip.gemm(hi, w, M=M, N=N, K=K, lda=K, out32=o_p, ldc32=N, a16_lo=lo)
        hip.gemm(hi, w, M=M, N=N, K=K, lda=K, out32=o_h, ldc32=N)
    finally:
        hip.set_option(hip.OPT_GEMM_TILE, prev)
    torch.cuda.synchronize()
    err_p, err_h = (o_p - ref).abs().max().item(), (o_h - ref).abs().max().item()
    print(f"precise {err_p:.3e}  fp16 {err_h:.3e}")
    assert err_h > 1e-4 and err_p < 2e-5 and err_p < err_h / 50
    e_p = torch.zeros(M, N, device=DEV)
    emu.gemm(hi, w, M=M, N=N, K=K, lda=K, out32=e_p, ldc32=N, a16_lo=lo)
    check("precise vs emu", o_p, e_p, 2e-5, 1e-5)


def test_gemm_precise_operand_conv_and_splitk():
    # conv3x3 (both K orders), conv1d and the split-K path with a lo plane
    for Cin in (24, 64):
        F, H, W, N = 2, 9, 12, 320
        M, K = F * H * W, 9 * Cin
        x32 = rnd(F, H, W, Cin, seed=3) * 2.0
        hi, lo = _split(x32)
        w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=4)
        conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
        h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
            a16=hi, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, out32=o["o"], ldc32=N, a16_lo=lo))
        check("conv3x3 precise", h["o"], e["o"], 2e-5, 1e-5)
    B, T, Npix, C = 1, 8, 40, 1280                # K = 3840: split-K regime
    M, N, K = B * T * Npix, C, 3 * C
    assert _splits(M=M, N=N, K=K, a_mode=hip.A_CONV1D_T) >= 2
    x32 = rnd(M, C, seed=5)
    hi, lo = _split(x32)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=6)
    res = rnd(M, N, seed=7)
    h, e = _run_both("gemm", lambda: dict(o=res.clone()), lambda o: dict(
        a16=hi, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=rnd(N, seed=8),
        res1=o["o"], ldr1=N, out32=o["o"], ldc32=N, a16_lo=lo))
    check("conv1d_t split-K precise", h["o"], e["o"], 3e-5, 1e-5)


def test_gemm_generic_epilogue_ragged_everything():
    # N % 8 != 0, odd leading dimensions, three added streams + V^T with ragged groups: the scalar fallback
    M, N, K = 333, 100, 72
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, rowb, r1, r2 = rnd(N), rnd(3, N), rnd(M, N + 1), rnd(M, N + 3)

    def outs():
        return dict(o32=torch.zeros(M, N + 5, device=DEV), o16=torch.zeros(M, N + 2, device=DEV, dtype=torch.float16),
                    lo=torch.zeros(M, N + 2, device=DEV, dtype=torch.float16))
    h, e = _run_both("gemm", outs, lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, rowbias=rowb, rb_rows=40, rb_mod=3, res1=r1, ldr1=N + 1,
        res2=r2, ldr2=N + 3, act=hip.ACT_SILU, out32=o["o32"], ldc32=N + 5, out16=o["o16"], ldc16=N + 2, out16_lo=o["lo"]))
    check("generic out32", h["o32"], e["o32"], 2e-3)
    check("generic out16", h["o16"], e["o16"], 6e-3)
    G, t_rows, C = 3, 75, 64                         # t_rows % 8 != 0 -> generic V^T
    M, N, K = G * t_rows, 3 * C, C
    a, w = rnd(M, K, dtype=torch.float16), rnd(N, K, scale=K ** -0.5, dtype=torch.float16)

    def outs2():
        return dict(qk=torch.zeros(M, 2 * C, device=DEV, dtype=torch.float16),
                    vt=torch.zeros(G, C, t_rows + 3, device=DEV, dtype=torch.float16))
    h, e = _run_both("gemm", outs2, lambda o: dict(
        a16=a, w16=w, M=M, N=N, K=K, lda=K, out16=o["qk"], ldc16=2 * C, out16t=o["vt"], ldt=t_rows + 3,
        t_rows=t_rows, t_gstride=C * (t_rows + 3), n_split=2 * C))
    check("generic qk", h["qk"], e["qk"], 4e-3)
    check("generic vt", h["vt"], e["vt"], 4e-3)


def test_gemm_rejects_a_struct_of_another_abi():
    import ctypes
    p = hip.GemmParams()
    assert hip.load().pnc_gemm_f16(ctypes.byref(p), None) == -3           # PNC_EABI
    p.struct_bytes = 224                                                   # round-1 layout
    assert hip.load().pnc_gemm_f16(ctypes.byref(p), None) == -3

@pytest.mark.parametrize("kind", ["plain", "res", "rowbias"])
@pytest.mark.parametrize("M,N,K,tile", [(1000, 320, 320, "256x320"), (777, 128, 192, "128x128"), (300, 64, 64, "128x128"),
                                        (600, 640, 128, None), (500, 320, 64, "256x128")])
def test_gemm_fused_layernorm(M, N, K, tile, kind):
    """LayerNorm of the fp32 output rows written by the GEMM (PncGemmParams.ln_*): fused into the epilogue when one
    workgroup owns whole rows (N = 320 on 256x320, N <= 128 on 128x128), the library's LayerNorm kernel after the GEMM
    otherwise (N = 640; a geometry without the variant) — same result, also in place on the residual stream."""
    a, w = rnd(M, K, dtype=torch.float16, seed=1), rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
    bias, res, rowb = rnd(N, seed=3), rnd(M, N, seed=4) * 2.0 + 0.7, rnd(4, N, seed=5)
    gamma, beta = rnd(N, seed=6) * 0.5 + 1, rnd(N, seed=7) * 0.3

    def outs():
        return dict(o32=res.clone(), ln=torch.zeros(M, N, device=DEV, dtype=torch.float16))

    def kw(o):
        d = dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, out32=o["o32"], ldc32=N, ln_gamma=gamma, ln_beta=beta,
                 ln_out16=o["ln"], ldln=N)
        if kind == "res":
            d.update(res1=o["o32"], ldr1=N)
        elif kind == "rowbias":
            d.update(rowbias=rowb, rb_rows=100, rb_mod=4)
        return d
    prev = hip.set_option(hip.OPT_GEMM_TILE, TILES[tile]) if tile else None
    try:
        h, e = _run_both("gemm", outs, kw)
    finally:
        if tile:
            hip.set_option(hip.OPT_GEMM_TILE, prev)
    check("out32", h["o32"], e["o32"], 2e-3)
    check("layernorm(out32)", h["ln"], e["ln"], 4e-3)
    # fused and un-fused (PNC_OPT_GEMM_FUSE_LN = 0: LayerNorm kernel after the GEMM) agree to fp16 rounding of the output
    prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, 0)
    try:
        u = outs()
        hip.gemm(**kw(u))
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_GEMM_FUSE_LN, prev)
    assert torch.equal(u["o32"], h["o32"])
    check("fused vs kernel", h["ln"], u["ln"], 2e-3, 1e-3)
    # ln_* handed to the library unconditionally (its own trailing LayerNorm launch where rows span workgroups)
    lb = outs()
    hip.gemm(**kw(lb), ln_in_library=True)
    torch.cuda.synchronize()
    assert torch.equal(lb["o32"], h["o32"])
    check("library-side ln", lb["ln"], h["ln"], 2e-3, 1e-3)
    # the normalised rows belong to the fp32 rows the SAME launch wrote
    ref = torch.nn.functional.layer_norm(h["o32"], (N,), gamma, beta, 1e-5)
    check("ln vs own out32", h["ln"], ref, 3e-3)


@pytest.mark.parametrize("M,N,K,geglu", [(1500, 2560, 128, True), (2000, 3840, 64, False), (700, 5120, 64, True)])
def test_gemm_grouped_tile_order_is_bit_identical(M, N, K, geglu):
    """PNC_OPT_GEMM_GROUP_M walks the output tiles in groups of row panels (L2 reuse of W): same tiles, same bits, for every
    group size incl. ones that do not divide the panel count, with and without the tail split."""
    a = rnd(M, K, dtype=torch.float16, seed=1)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
    bias = rnd(N, seed=3)
    No = N // 2 if geglu else N

    def run(g):
        o = torch.zeros(M, No, device=DEV, dtype=torch.float16)
        prev = hip.set_option(hip.OPT_GEMM_GROUP_M, g)
        try:
            hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=geglu, out16=o, ldc16=No)
            torch.cuda.synchronize()
        finally:
            hip.set_option(hip.OPT_GEMM_GROUP_M, prev)
        return o
    ref = run(1)                                     # 1 = plain order
    e = torch.zeros(M, No, device=DEV, dtype=torch.float16)
    emu.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=geglu, out16=e, ldc16=No)
    check("plain order vs emu", ref, e, 4e-3)
    for g in (0, 2, 3, 4, 7):
        assert torch.equal(run(g), ref), g


def _with_stencil_tiles(opt, fn):
    prev = hip.set_option(hip.OPT_STENCIL_TILES, opt)
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_STENCIL_TILES, prev)


@pytest.mark.parametrize("F,H,W,Cin,N,epi", [
    (2, 16, 32, 64, 320, "o32"),        # 16x16 tiles, BN = 320, one slice
    (1, 32, 48, 192, 640, "o32"),       # three slices (both halo buffers recycled), two column tiles
    (3, 8, 64, 128, 256, "o32+o16"),    # 8x32 tiles, BN = 256
    (1, 16, 16, 320, 384, "r1"),        # one tile per frame: every halo edge is padding; N = 384 = 256 + half a tile
    (2, 48, 16, 128, 960, "r1+o16"),    # tall image, three 320-column tiles
    (1, 8, 32, 64, 64, "silu16"),       # a narrow conv (forced geometry only)
    (2, 16, 16, 128, 320, "o32+lo"),    # precise operand: the lo plane's pass, the 2^-11 scale, the hi plane's pass
    (12, 64, 96, 64, 320, "r1"),        # 288 tiles: a full round + 32 tiles x 4 quarter workgroups
    (5, 32, 64, 64, 640, "o32"),        # 40 x 2 = 80 tiles: every tile as 2 half-tile workgroups
])
def test_gemm_conv3x3_stencil_tiles_are_bit_identical(F, H, W, Cin, N, epi):
    """PNC_OPT_STENCIL_TILES: the spatial-tile kernel (one halo of the input per 64-channel slice, nine taps read from it)
    computes the same products in the same K order as the per-tap gather: bit-identical for every fast epilogue and
    with a precise operand; and the per-tap gather agrees with the emulation."""
    M, K = F * H * W, 9 * Cin
    x32 = rnd(F, H, W, Cin, seed=11)
    x, lo = _split(x32) if "lo" in epi else (x32.half(), None)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=12)
    bias = rnd(N, seed=13)
    res = rnd(M, N, seed=14)
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)

    def outs():
        return dict(o32=res.clone() if "r1" in epi else torch.zeros(M, N, device=DEV),
                    o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))

    def kw(o):
        k = dict(a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias, a16_lo=lo)
        if epi == "silu16":
            k.update(act=hip.ACT_SILU, out16=o["o16"], ldc16=N)
        else:
            k.update(out32=o["o32"], ldc32=N)
            if "o16" in epi:
                k.update(out16=o["o16"], ldc16=N)
            if "r1" in epi:
                k.update(res1=o["o32"], ldr1=N)
        return k
    plain, tiles, e = outs(), outs(), outs()
    _with_stencil_tiles(0, lambda: hip.gemm(**kw(plain)))
    emu.gemm(**kw(e))
    tol = (2e-5, 1e-5) if lo is not None else (2e-3, 2e-3)
    check("per-tap gather vs emu (fp32)", plain["o32"], e["o32"], *tol)
    check("per-tap gather vs emu (fp16)", plain["o16"], e["o16"], 4e-3)
    # small grids are one sparse round: with the tail split every tile runs as 4 quarter-tile workgroups, without as one;
    # + 4: the fragment addresses next to the reads (round 5) instead of one MFMA batch ahead of them (round 6's default)
    for tail, forced in ((1, 2), (0, 2), (1, 6)):
        tiles = outs()
        prev = hip.set_option(hip.OPT_GEMM_TAIL_SPLIT, tail)
        try:
            _with_stencil_tiles(forced, lambda: hip.gemm(**kw(tiles)))
        finally:
            hip.set_option(hip.OPT_GEMM_TAIL_SPLIT, prev)
        assert torch.equal(tiles["o32"], plain["o32"]), (tail, (tiles["o32"] - plain["o32"]).abs().max().item())
        assert torch.equal(tiles["o16"], plain["o16"]), tail


# ---------------------------------------------------------------------------------------- split K
def _splits(**kw):
    """K slices the library would run for this problem (0 workspace -> 1 slice)."""
    import ctypes
    p = hip.GemmParams()
    p.struct_bytes = ctypes.sizeof(hip.GemmParams)
    for k, v in kw.items():
        setattr(p, k, v)
    return hip.load().pnc_gemm_workspace_floats(ctypes.byref(p)) // (kw["M"] * kw["N"])


@pytest.mark.parametrize("M,N,K", [(300, 256, 3136), (3072, 1280, 5120), (1000, 512, 6400)])
def test_gemm_splitk_full_epilogue_and_determinism(M, N, K):
    # small-M / long-K: K is cut into slices whose fp32 partials are summed in a fixed order by a second launch
    assert _splits(M=M, N=N, K=K) >= 2
    a = rnd(M, K, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, rowb = rnd(N), rnd(4, N)
    res1_init, res2 = rnd(M, N + 8), rnd(M, N)

    def outs():
        return dict(o32=res1_init.clone(), o16=torch.zeros(M, N, device=DEV, dtype=torch.float16))
    kw = lambda o: dict(a16=a, w16=w, M=M, N=N, K=K, lda=K, bias=bias, rowbias=rowb, rb_rows=50, rb_mod=4,
                        res1=o["o32"], ldr1=N + 8, res2=res2, ldr2=N, out32=o["o32"], ldc32=N + 8,
                        out16=o["o16"], ldc16=N)
    h, e = _run_both("gemm", outs, kw)
    check("out32", h["o32"], e["o32"], 2e-3)
    check("out16", h["o16"], e["o16"], 6e-3)
    h2 = outs()
    hip.gemm(**kw(h2))
    torch.cuda.synchronize()
    assert torch.equal(h2["o32"], h["o32"]) and torch.equal(h2["o16"], h["o16"])


def test_gemm_splitk_conv3x3_and_conv1d():
    F, H, W, Cin, N = 2, 8, 24, 384, 256          # K = 3456 -> 54 K tiles, 2 x 1 output tiles
    M, K = F * H * W, 9 * Cin
    assert _splits(M=M, N=N, K=K, a_mode=hip.A_CONV3X3) >= 2
    x = rnd(F, H, W, Cin, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    bias, emb = rnd(N), rnd(F, N)
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
    h, e = _run_both("gemm", lambda: dict(o=torch.zeros(M, N, device=DEV)), lambda o: dict(
        a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias, rowbias=emb, rb_rows=H * W, rb_mod=F,
        out32=o["o"], ldc32=N))
    check("conv3x3 split-K", h["o"], e["o"], 2e-3)
    B, T, Npix, C = 1, 8, 40, 1280                # K = 3840
    M, N, K = B * T * Npix, C, 3 * C
    assert _splits(M=M, N=N, K=K, a_mode=hip.A_CONV1D_T) >= 2
    x = rnd(M, C, dtype=torch.float16)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    res = rnd(M, N)
    h, e = _run_both("gemm", lambda: dict(o=res.clone()), lambda o: dict(
        a16=x, w16=w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=rnd(N, seed=5),
        res1=o["o"], ldr1=N, out32=o["o"], ldc32=N))
    check("conv1d_t split-K", h["o"], e["o"], 2e-3)


# -------------------------------------------------------------------------------------- attention
INTRA = [[0], [1], [2], [3], [4], [5]]
CROSS = [[5, 1], [0, 2], [1, 3], [2, 4], [3, 5], [4]]      # view 5 sees view 4 only (reference quirk)


def _qkv(G, N, C, seed):
    q = rnd(G * N, C, dtype=torch.float16, seed=seed)
    k = rnd(G * N, C, dtype=torch.float16, seed=seed + 1)
    v = rnd(G * N, C, dtype=torch.float16, seed=seed + 2)
    vt = v.view(G, N, C).permute(0, 2, 1).contiguous()
    return q, k, v, vt


@pytest.mark.parametrize("G,H,W,heads,segs", [
    (2, 8, 96, 1, INTRA), (2, 8, 96, 2, CROSS), (1, 4, 48, 2, INTRA), (1, 4, 48, 1, CROSS),
    (1, 32, 384, 1, CROSS), (2, 16, 16, 1, [[0]]), (1, 16, 192, 2, INTRA),
    (2, 2, 24, 1, CROSS), (1, 1, 12, 2, INTRA), (2, 3, 36, 1, CROSS), (1, 5, 10, 1, [[0]]),
    (1, 24, 150, 1, CROSS), (1, 17, 120, 2, INTRA)])
def test_attn_views_self(G, H, W, heads, segs):
    C, N, views = heads * 64, H * W, len(segs)
    q, k, _, vt = _qkv(G, N, C, 5)
    kw = dict(groups=G, heads=heads, H=H, W=W, views=views, kvH=H, kvW=W, kv_views=views,
              kv_rows_per_group=N, q_per_kv=1, kv_valid=H * (W // views), segs=segs, scale=0.125)
    oh = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    hip.attn_views(q, C, k, C, vt, N, C * N, oh, C, **kw)
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, **kw)
    torch.cuda.synchronize()
    check("attn_views", oh, oe, 3e-3)


@pytest.mark.parametrize("variant", ["41", "81", "42", "82"])
@pytest.mark.parametrize("G,H,W,heads,segs", [(1, 24, 150, 1, CROSS), (2, 8, 96, 2, CROSS), (1, 3, 36, 1, INTRA)])
def test_attn_views_every_variant(variant, G, H, W, heads, segs):
    # (waves, query blocks per wave) variants of the kernel, forced regardless of the size heuristic
    prev = hip.set_option(hip.OPT_ATTN_VARIANT, int(variant))
    try:
        test_attn_views_self(G, H, W, heads, segs)
    finally:
        hip.set_option(hip.OPT_ATTN_VARIANT, prev)


@pytest.mark.parametrize("variant", [0, 41, 81, 42, 82])
@pytest.mark.parametrize("G,H,W,heads,segs", [(2, 8, 96, 2, CROSS), (1, 16, 192, 1, CROSS), (1, 32, 384, 1, INTRA), (1, 32, 384, 1, CROSS)])
def test_attn_views_incremental_tile_addresses_are_bit_identical(variant, G, H, W, heads, segs):
    """PNC_OPT_ATTN_DMA 1 (lane constant + wave-uniform tile offset, round 4) against 2 (key -> (row, column) split recomputed per
    tile): the same addresses, hence the same bits — views of 16 / 32 / 64 columns (the network's levels 2 / 1 / 0), one and two
    key segments, 2 .. 64 tiles per view; and against the emulation."""
    C, N, views = heads * 64, H * W, len(segs)
    q, k, _, vt = _qkv(G, N, C, 7)
    kw = dict(groups=G, heads=heads, H=H, W=W, views=views, kvH=H, kvW=W, kv_views=views,
              kv_rows_per_group=N, q_per_kv=1, kv_valid=H * (W // views), segs=segs, scale=0.125)
    outs = []
    pv = hip.set_option(hip.OPT_ATTN_VARIANT, variant)
    try:
        for mode in (1, 2):
            pm = hip.set_option(hip.OPT_ATTN_DMA, mode)
            o = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
            hip.attn_views(q, C, k, C, vt, N, C * N, o, C, **kw)
            hip.set_option(hip.OPT_ATTN_DMA, pm)
            outs.append(o)
    finally:
        hip.set_option(hip.OPT_ATTN_VARIANT, pv)
    oe = torch.zeros_like(outs[0])
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, **kw)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    check("attn_views_inc", outs[0], oe, 3e-3)


@pytest.mark.parametrize("scale_q", [1.0, 4.0, 12.0])
def test_attn_views_deferred_running_max(scale_q):
    """PNC_OPT_ATTN_DEFER_MAX: the running maximum is only raised (and the accumulators rescaled) when some query of the wave
    exceeds it by more than k in the exp2 domain; k = 0 is the classic online softmax.  Same softmax for every k — checked on
    near-uniform, sharp and very sharp (one-hot-like: growth of the max by far more than k between tiles) score rows."""
    G, H, W, heads = 1, 16, 192, 2
    C, N = heads * 64, H * W
    q, k, _, vt = _qkv(G, N, C, 13)
    q = q * scale_q
    kw = dict(groups=G, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W, kv_views=6, kv_rows_per_group=N,
              q_per_kv=1, kv_valid=H * (W // 6), segs=CROSS, scale=0.125)
    oe = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, **kw)
    for thr in (0, 8, 14):
        prev = hip.set_option(hip.OPT_ATTN_DEFER_MAX, thr)
        try:
            oh = torch.zeros_like(oe)
            hip.attn_views(q, C, k, C, vt, N, C * N, oh, C, **kw)
            torch.cuda.synchronize()
        finally:
            hip.set_option(hip.OPT_ATTN_DEFER_MAX, prev)
        check(f"attn_views_defer{thr}_q{scale_q}", oh, oe, 5e-3)


@pytest.mark.parametrize("variant", [42, 82, 81, 41])
@pytest.mark.parametrize("scale_q,spike", [(1.0, 0.0), (4.0, 0.0), (12.0, 0.0), (1.0, 40.0), (1.0, 400.0)])
def test_attn_views_sum_triggered_running_max(variant, scale_q, spike):
    """PNC_OPT_ATTN_SUM_TRIGGER (round 6): after the first K/V tile the probabilities are formed against the running maximum as it is,
    and only a tile whose row sum reaches 2^k (or is not finite) takes the exact path (row maximum, rescale).  Same softmax for every
    k, 0 = the row maximum of every tile — on near-uniform, sharp and very sharp rows, and with ONE key row spiked in a LATE tile so
    that its raw score exceeds everything before it by far more than any fp16 / fp32 exponent range allows to defer (the optimistic
    probabilities overflow: the not-finite sum must trigger the redo).  Every output finite; all k agree with the emulation."""
    G, H, W, heads = 1, 16, 192, 2
    C, N = heads * 64, H * W
    q, k, _, vt = _qkv(G, N, C, 13)
    q = q * scale_q
    if spike:
        # key (y = 9, x = 7) of every view: tile 4 of the 8 tiles of a 512-key view; aligned with query (0, 0)'s direction
        k4 = k.view(G, H, W, C)
        for v in range(6):
            k4[0, 9, v * 32 + 7] = q.view(G, H, W, C)[0, 0, v * 32] * spike
    kw = dict(groups=G, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W, kv_views=6, kv_rows_per_group=N,
              q_per_kv=1, kv_valid=H * (W // 6), segs=CROSS, scale=0.125)
    oe = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, **kw)
    pv = hip.set_option(hip.OPT_ATTN_VARIANT, variant)
    outs = []
    try:
        for trig in (0, 4, 12, 14):
            prev = hip.set_option(hip.OPT_ATTN_SUM_TRIGGER, trig)
            try:
                oh = torch.zeros_like(oe)
                hip.attn_views(q, C, k, C, vt, N, C * N, oh, C, **kw)
                torch.cuda.synchronize()
            finally:
                hip.set_option(hip.OPT_ATTN_SUM_TRIGGER, prev)
            assert torch.isfinite(oh.float()).all(), (trig, "non-finite attention output")
            check(f"attn_views_sumtrig{trig}_v{variant}_q{scale_q}_s{spike}", oh, oe, 5e-3)
            outs.append(oh)
    finally:
        hip.set_option(hip.OPT_ATTN_VARIANT, pv)


@pytest.mark.parametrize("variant,dma", [(0, 1), (42, 1), (42, 2), (82, 1), (81, 2), (41, 1), (0, 0)])
@pytest.mark.parametrize("n_local,H,Wv,heads", [(3, 16, 32, 2), (2, 8, 16, 1), (1, 32, 64, 1)])
def test_attn_views_halo_views_of_a_view_band(variant, dma, n_local, H, Wv, heads):
    """PncAttnParams.k_halo / vt_halo (round 5; kernel-level test asked for by ADVICE r5): a rank's band of n_local views attends
    its circular neighbours' edge views, which live in view column 0 of two buffers with the BAND's geometry (segment ids -1 and
    kv_views) — against the emulation AND against the same attention over the explicitly concatenated n_local + 2 views (round 3-4
    form), for the incremental and the recomputed tile addresses, LDS-DMA and register staging, every workgroup shape.  The halo
    buffers' other columns hold NaN: nothing outside view column 0 may be read."""
    G = 2
    W = n_local * Wv
    C, N = heads * 64, H * W
    q, k, _, vt = _qkv(G, N, C, 21)
    nan = float("nan")
    kh = [torch.full((G * N, C), nan, device=DEV, dtype=torch.float16) for _ in range(2)]
    vh = [torch.full((G, C, N), nan, device=DEV, dtype=torch.float16) for _ in range(2)]
    kn = [rnd(G, H, Wv, C, dtype=torch.float16, seed=31 + i) for i in range(2)]
    vn = [rnd(G, C, H, Wv, dtype=torch.float16, seed=41 + i) for i in range(2)]
    for i in range(2):
        kh[i].view(G, H, W, C)[:, :, :Wv] = kn[i]
        vh[i].view(G, C, H, W)[..., :Wv] = vn[i]
    # every local view attends its left and right neighbour (the panorama's INTER_SEGS restricted to an interior band)
    segs = [[(v - 1) if v > 0 else -1, (v + 1) if v < n_local - 1 else n_local] for v in range(n_local)]
    kw = dict(groups=G, heads=heads, H=H, W=W, views=n_local, kvH=H, kvW=W, kv_views=n_local, kv_rows_per_group=N, q_per_kv=1,
              kv_valid=H * Wv, segs=segs, scale=0.125)
    oe = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, k_halo=kh, vt_halo=vh, **kw)
    # the explicit form: [left | band | right] concatenated along the width
    We = W + 2 * Wv
    ke = torch.cat([kn[0], k.view(G, H, W, C), kn[1]], dim=2).contiguous().view(G * H * We, C)
    ve = torch.cat([vn[0], vt.view(G, C, H, W), vn[1]], dim=3).contiguous().view(G, C, H * We)
    kw2 = dict(kw, kvW=We, kv_views=n_local + 2, kv_rows_per_group=H * We, segs=[[u + 1 for u in row] for row in segs])
    ox = torch.zeros_like(oe)
    pv, pd = hip.set_option(hip.OPT_ATTN_VARIANT, variant), hip.set_option(hip.OPT_ATTN_DMA, dma)
    try:
        oh = torch.zeros_like(oe)
        hip.attn_views(q, C, k, C, vt, N, C * N, oh, C, k_halo=kh, vt_halo=vh, **kw)
        hip.attn_views(q, C, ke, C, ve, H * We, C * H * We, ox, C, **kw2)
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_ATTN_VARIANT, pv)
        hip.set_option(hip.OPT_ATTN_DMA, pd)
    assert torch.isfinite(oh.float()).all()
    check("attn_views_halo", oh, oe, 3e-3)
    assert torch.equal(oh, ox)              # same keys in the same tile order: the same bits as the concatenated form


def test_groupnorm_combine_kernel():
    """pnc_groupnorm_combine (round 4): Chan combination of the all-gathered chunk records of a view group's bands, against the
    float64 formula; the apply kernel fed with the combined records normalises with the statistics of the concatenation."""
    F, Npix, C, ppc, parts = 2, 300, 320, 64, 3
    nchunk = (Npix + ppc - 1) // ppc
    xs = [rnd(F * Npix, C, seed=30 + s) * (1.0 + 0.5 * s) + 0.3 * s for s in range(parts)]
    recs = torch.zeros(parts, F * nchunk * 96, device=DEV)
    for s in range(parts):
        hip.groupnorm_stats(xs[s], C, F, Npix, C, ppc, recs[s])
    out_h, out_e = torch.full((F * nchunk * 96,), -1.0, device=DEV), torch.zeros(F * nchunk * 96, device=DEV)
    hip.groupnorm_combine(recs.view(-1), parts, F, nchunk, out_h)
    emu.groupnorm_combine(recs.view(-1), parts, F, nchunk, out_e)
    torch.cuda.synchronize()
    oh, oe = out_h.view(F, nchunk, 32, 3), out_e.view(F, nchunk, 32, 3)
    assert not oh[:, 1:].any() and torch.equal(oh[:, 0, :, 0], oe[:, 0, :, 0])
    assert torch.allclose(oh[:, 0, :, 1], oe[:, 0, :, 1], atol=1e-6) and torch.allclose(oh[:, 0, :, 2], oe[:, 0, :, 2], rtol=1e-5)
    # band 0 normalised with the panorama's statistics == GroupNorm over the concatenated bands, band 0's rows
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    y = torch.zeros(F * Npix, C, device=DEV, dtype=torch.float16)
    hip.groupnorm_apply(xs[0], C, F, Npix, C, ppc, out_h, gamma, beta, 1e-5, 0, y, C)
    pano = torch.cat([x.view(F, Npix, C) for x in xs], dim=1)
    ref = torch.nn.functional.group_norm(pano.permute(0, 2, 1), 32, gamma, beta, 1e-5)[:, :, :Npix]
    check("groupnorm_combined", y.view(F, Npix, C).permute(0, 2, 1), ref, 4e-3)


@pytest.mark.parametrize("G,L,Lp,heads", [(2, 77, 80, 16), (1, 200, 200, 2), (3, 33, 40, 1)])
def test_attn_views_causal_text_tower(G, L, Lp, heads):
    """PncAttnParams.causal (round 4): query i attends keys j <= i — the OpenCLIP text tower's mask, all heads of all prompts in one
    launch (a prompt = a group of one view with Lp padded rows, L valid keys)"""
    C = heads * 64
    qk = rnd(G * Lp, 2 * C, dtype=torch.float16, seed=51)
    v = rnd(G * Lp, C, dtype=torch.float16, seed=52)
    vt = v.view(G, Lp, C).permute(0, 2, 1).contiguous()
    kw = dict(groups=G, heads=heads, H=1, W=Lp, views=1, kvH=1, kvW=Lp, kv_views=1, kv_rows_per_group=Lp, q_per_kv=1,
              kv_valid=L, segs=[[0]], scale=0.125, causal=True)
    oh = torch.zeros(G * Lp, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    hip.attn_views(qk, 2 * C, qk[:, C:], 2 * C, vt, Lp, C * Lp, oh, C, **kw)
    emu.attn_views(qk, 2 * C, qk.reshape(-1)[C:], 2 * C, vt, Lp, C * Lp, oe, C, **kw)
    torch.cuda.synchronize()
    check("attn_views_causal", oh, oe, 3e-3)
    # row 0 of every prompt attends key 0 only: its output is v[0]
    assert torch.equal(oh.view(G, Lp, C)[:, 0], v.view(G, Lp, C)[:, 0])


def test_attn_views_sharp_softmax():
    # large-magnitude scores: exercises the running-max rescale across KV tiles
    G, H, W, heads = 1, 8, 96, 1
    C, N = 64, H * W
    q, k, _, vt = _qkv(G, N, C, 9)
    q = q * 4
    kw = dict(groups=G, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W, kv_views=6, kv_rows_per_group=N,
              q_per_kv=1, kv_valid=H * (W // 6), segs=CROSS, scale=0.125)
    oh = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    hip.attn_views(q, C, k, C, vt, N, C * N, oh, C, **kw)
    emu.attn_views(q, C, k, C, vt, N, C * N, oe, C, **kw)
    torch.cuda.synchronize()
    check("attn_views_sharp", oh, oe, 5e-3)


@pytest.mark.parametrize("B,T,H,W,heads", [(2, 2, 8, 96, 2), (1, 8, 4, 48, 1)])
def test_attn_views_text(B, T, H, W, heads):
    # 77 text keys per sample, padded to 80 rows, shared by the T frames of the sample
    C, N, G = heads * 64, H * W, B * T
    q = rnd(G * N, C, dtype=torch.float16, seed=3)
    k = rnd(B * 80, C, dtype=torch.float16, seed=4)
    v = rnd(B * 80, C, dtype=torch.float16, seed=6)
    k.view(B, 80, C)[:, 77:] = 0
    v.view(B, 80, C)[:, 77:] = 0
    vt = v.view(B, 80, C).permute(0, 2, 1).contiguous()
    kw = dict(groups=G, heads=heads, H=H, W=W, views=1, kvH=1, kvW=80, kv_views=1, kv_rows_per_group=80,
              q_per_kv=T, kv_valid=77, segs=[[0]], scale=0.125)
    oh = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    hip.attn_views(q, C, k, C, vt, 80, C * 80, oh, C, **kw)
    emu.attn_views(q, C, k, C, vt, 80, C * 80, oe, C, **kw)
    torch.cuda.synchronize()
    check("attn_text", oh, oe, 3e-3)


@pytest.mark.parametrize("B,T,H,W,heads,n_text,sharp", [(2, 2, 8, 96, 2, 77, 1.0), (1, 8, 4, 48, 1, 77, 1.0), (2, 4, 16, 48, 5, 77, 6.0),
                                                        (1, 2, 13, 31, 7, 65, 1.0), (2, 1, 8, 96, 12, 96, 3.0), (1, 3, 8, 24, 20, 80, 1.0)])
def test_attn_text_single_pass_kernel(B, T, H, W, heads, n_text, sharp):
    """attn_text_kernel (round 6; PNC_OPT_ATTN_VARIANT = 43 forces it on grids the default leaves to attn_views_kernel): 64 < kv_valid
    <= 96 keys shared by the queries of a sample, 128 queries x 5 heads per workgroup, single-pass softmax, padding keys masked
    through the accumulator's initial value.  Against the emulation and against attn_views_kernel on the same operands — ragged query
    counts (Nq % 128 != 0), head counts that are no multiple of 5, 65 / 77 / 80 / 96 valid keys, sharp rows; padding key rows hold
    NaN-free garbage that must not reach the output."""
    C, N, G = heads * 64, H * W, B * T
    q = rnd(G * N, C, dtype=torch.float16, seed=3) * sharp
    k = rnd(B * 96, C, dtype=torch.float16, seed=4)
    v = rnd(B * 96, C, dtype=torch.float16, seed=6)
    k.view(B, 96, C)[:, n_text:] = 0                    # (the product zero-pads: engine.Runtime.set_context)
    v.view(B, 96, C)[:, n_text:] = 7.0                  # values of padding keys must not matter: their probabilities are exact zeros
    vt = v.view(B, 96, C).permute(0, 2, 1).contiguous()
    kw = dict(groups=G, heads=heads, H=H, W=W, views=1, kvH=1, kvW=96, kv_views=1, kv_rows_per_group=96,
              q_per_kv=T, kv_valid=n_text, segs=[[0]], scale=0.125)
    oe = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    emu.attn_views(q, C, k, C, vt, 96, C * 96, oe, C, **kw)
    outs = {}
    for variant in (43, 42):
        prev = hip.set_option(hip.OPT_ATTN_VARIANT, variant)
        try:
            o = torch.full((G * N, C), float("nan"), device=DEV, dtype=torch.float16)
            hip.attn_views(q, C, k, C, vt, 96, C * 96, o, C, **kw)
            torch.cuda.synchronize()
        finally:
            hip.set_option(hip.OPT_ATTN_VARIANT, prev)
        assert torch.isfinite(o.float()).all(), variant
        check(f"attn_text_v{variant}", o, oe, 3e-3)
        outs[variant] = o
    assert (outs[43].float() - outs[42].float()).abs().max().item() <= 4e-3


def test_attn_views_strided_qkv_buffer():
    # q and k live in one [M, 2C] projection buffer (ld = 2C), as the engine lays them out
    G, H, W, heads = 2, 8, 96, 2
    C, N = heads * 64, H * W
    qk = rnd(G * N, 2 * C, dtype=torch.float16, seed=11)
    _, _, _, vt = _qkv(G, N, C, 12)
    kw = dict(groups=G, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W, kv_views=6, kv_rows_per_group=N,
              q_per_kv=1, kv_valid=H * (W // 6), segs=INTRA, scale=0.125)
    oh = torch.zeros(G * N, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    hip.attn_views(qk, 2 * C, qk[:, C:], 2 * C, vt, N, C * N, oh, C, **kw)
    emu.attn_views(qk, 2 * C, qk.reshape(-1)[C:], 2 * C, vt, N, C * N, oe, C, **kw)
    torch.cuda.synchronize()
    check("attn_views_strided", oh, oe, 3e-3)


@pytest.mark.parametrize("B,T,Npix,heads", [(2, 8, 96, 2), (1, 2, 50, 1), (2, 1, 64, 1), (1, 3, 33, 2)])
def test_attn_temporal(B, T, Npix, heads):
    C, M = heads * 64, B * T * Npix
    qkv = rnd(M, 3 * C, dtype=torch.float16, seed=21)
    oh = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    oe = torch.zeros_like(oh)
    flat = qkv.reshape(-1)
    hip.attn_temporal(qkv, 3 * C, qkv[:, C:], 3 * C, qkv[:, 2 * C:], 3 * C, oh, C,
                      B=B, T=T, Npix=Npix, heads=heads, scale=0.125)
    emu.attn_temporal(flat, 3 * C, flat[C:], 3 * C, flat[2 * C:], 3 * C, oe, C,
                      B=B, T=T, Npix=Npix, heads=heads, scale=0.125)
    torch.cuda.synchronize()
    check("attn_temporal", oh, oe, 3e-3)


# ------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("F,Npix,C,ppc,silu", [(2, 768, 64, 128, 1), (3, 500, 320, 128, 0),
                                               (1, 192, 1280, 64, 1), (2, 300, 1920, 128, 1), (1, 64, 2560, 16, 0)])
def test_groupnorm_spatial(F, Npix, C, ppc, silu):
    x = rnd(F * Npix, C) * 1.7 + 0.9
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    nchunk = (Npix + ppc - 1) // ppc
    ph = torch.zeros(F * nchunk * 32 * 3, device=DEV)
    yh = torch.zeros(F * Npix, C, device=DEV, dtype=torch.float16)
    hip.groupnorm_stats(x, C, F, Npix, C, ppc, ph)
    hip.groupnorm_apply(x, C, F, Npix, C, ppc, ph, gamma, beta, 1e-5, silu, yh, C)
    torch.cuda.synchronize()
    ref = torch.nn.functional.group_norm(x.view(F, Npix, C).permute(0, 2, 1), 32, gamma, beta, 1e-5)
    if silu:
        ref = torch.nn.functional.silu(ref)
    check("groupnorm", yh.view(F, Npix, C).permute(0, 2, 1), ref, 4e-3)


@pytest.mark.parametrize("B,T,Npix,C", [(2, 8, 96, 64), (1, 2, 77, 320), (2, 1, 64, 128), (1, 8, 40, 1280)])
def test_groupnorm_temporal(B, T, Npix, C):
    x = rnd(B * T * Npix, C) * 1.3 - 0.4
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    yh = torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.float16)
    ye = torch.zeros_like(yh)
    hip.groupnorm_temporal_silu(x, B, T, Npix, C, gamma, beta, 1e-5, yh)
    emu.groupnorm_temporal_silu(x, B, T, Npix, C, gamma, beta, 1e-5, ye)
    torch.cuda.synchronize()
    check("gn_temporal", yh, ye, 4e-3)


@pytest.mark.parametrize("M,C", [(1000, 64), (513, 320), (256, 1280), (7, 640)])
def test_layernorm(M, C):
    x = rnd(M, C) * 2.1 + 0.5
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    yh = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    hip.layernorm(x, C, M, C, gamma, beta, 1e-5, yh, C)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5)
    check("layernorm", yh, ref, 4e-3)


# ---------------------------------------------------------------------------------------- helpers
def test_linear_smallm_and_timestep_embedding():
    F, dim, K, N = 16, 320, 320, 1280
    t = torch.tensor([999, 959, 0, 1, 500, 17, 333, 666] * 2, device=DEV, dtype=torch.int64)
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half).to(DEV)
    eh = torch.zeros(F, dim, device=DEV)
    hip.timestep_embedding(t, F, dim, freqs, eh)
    args = t[:, None].float().cpu() * freqs.cpu()[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    torch.cuda.synchronize()
    check("timestep_embedding", eh.cpu(), ref, 2e-5, 0)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16)
    b = rnd(N)
    for M, si, so in [(16, False, True), (2, True, False), (5, False, False)]:
        oh, oe = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
        hip.linear_smallm(eh, dim, w, b, oh, N, M, N, K, si, so)
        emu.linear_smallm(eh, dim, w, b, oe, N, M, N, K, si, so)
        torch.cuda.synchronize()
        check("linear_smallm", oh, oe, 1e-4, 1e-4)


def test_layout_helpers():
    F, C1, C2, H, W, Cpad = 2, 4, 4, 8, 12, 8
    a, b = rnd(F, C1, H, W), rnd(F, C2, H, W)
    oh = torch.zeros(F * H * W, Cpad, device=DEV, dtype=torch.float16)
    hip.nchw_to_tokens_f16(a, C1, b, C2, F, H * W, Cpad, oh)
    ref = torch.cat([a, b], 1).permute(0, 2, 3, 1).reshape(F * H * W, Cpad).half()
    torch.cuda.synchronize()
    assert torch.equal(oh, ref)
    a19 = rnd(F, 19, H, W)
    o24 = torch.full((F * H * W, 24), 7.0, device=DEV, dtype=torch.float16)
    hip.nchw_to_tokens_f16(a19, 19, None, 0, F, H * W, 24, o24)
    torch.cuda.synchronize()
    assert torch.equal(o24[:, :19], a19.permute(0, 2, 3, 1).reshape(-1, 19).half()) and (o24[:, 19:] == 0).all()
    x = rnd(F * H * W, 32)
    back = torch.zeros(F, 4, H, W, device=DEV)
    hip.tokens_to_nchw_f32(x, 32, F, H * W, 4, back)
    torch.cuda.synchronize()
    assert torch.equal(back, x[:, :4].reshape(F, H, W, 4).permute(0, 3, 1, 2))
    M, Ca, Cb = 300, 128, 64
    aa, ss, cc = rnd(M, Ca), rnd(M, Cb), rnd(M, Cb)
    o32 = torch.zeros(M, Ca + Cb, device=DEV)
    o16 = torch.zeros(M, Ca + Cb, device=DEV, dtype=torch.float16)
    hip.concat_add(aa, Ca, ss, cc, Cb, M, o32, o16)
    torch.cuda.synchronize()
    ref = torch.cat([aa, ss + cc], 1)
    assert torch.equal(o32, ref) and torch.equal(o16, ref.half())
    y32, y16 = torch.zeros(M, Ca, device=DEV), torch.zeros(M, Ca, device=DEV, dtype=torch.float16)
    hip.add_f32(aa, aa * 2, M * Ca, y32, y16)
    torch.cuda.synchronize()
    assert torch.equal(y32, aa + aa * 2) and torch.equal(y16, (aa + aa * 2).half())
    hip.add_f32(aa, ss[:, :1].expand(M, Ca).contiguous(), M * Ca, aa, None)     # in place
    hip.cast_f16(aa, M * Ca, y16)
    torch.cuda.synchronize()
    assert torch.equal(y16, aa.half())


@pytest.mark.parametrize("F,Npix,C1,C2,ctrl,lo,ppc", [(2, 192, 320, 320, True, "e4m3", 64), (3, 100, 640, 320, False, None, 16), (1, 768, 1280, 1280, True, "f16", 16),
                                                     (2, 64, 64, 64, True, None, 64), (2, 3072, 1280, 640, True, "e4m3", 64), (2, 256, 640, 320, True, "e4m3", 128)])
def test_concat_add_with_groupnorm_records(F, Npix, C1, C2, ctrl, lo, ppc):
    """pnc_concat_add_stats (ABI 5): the concat's values equal pnc_concat_add's bit for bit, its records equal
    pnc_groupnorm_stats(..., 64) of the result up to fp32 summation order, they are reproducible bit for bit, and
    pnc_groupnorm_apply on them equals the two-launch path to fp16 rounding"""
    M, C = F * Npix, C1 + C2
    a, s_, c = rnd(M, C1, seed=1) * 2 + 0.3, rnd(M, C2, seed=2), (rnd(M, C2, seed=3) if ctrl else None)
    lod = {None: None, "e4m3": torch.uint8, "f16": torch.float16}[lo]

    def outs():
        return dict(o32=torch.zeros(M, C, device=DEV), o16=torch.zeros(M, C, device=DEV, dtype=torch.float16),
                    lo=None if lod is None else torch.zeros(M, C, device=DEV, dtype=lod))
    nrec = -(-Npix // ppc)
    ref, got, got2 = outs(), outs(), outs()
    part, part2 = torch.zeros(F * nrec * 96, device=DEV), torch.zeros(F * nrec * 96, device=DEV)
    hip.concat_add(a, C1, s_, c, C2, M, ref["o32"], ref["o16"], ref["lo"])
    hip.concat_add(a, C1, s_, c, C2, M, got["o32"], got["o16"], got["lo"], gn_part=part, frames=F, ppc=ppc)
    hip.concat_add(a, C1, s_, c, C2, M, got2["o32"], got2["o16"], got2["lo"], gn_part=part2, frames=F, ppc=ppc)
    pref = torch.zeros(F * nrec * 96, device=DEV)
    hip.groupnorm_stats(ref["o32"], C, F, Npix, C, ppc, pref)
    torch.cuda.synchronize()
    assert torch.equal(got["o32"], ref["o32"]) and torch.equal(got["o16"], ref["o16"]) and torch.equal(part, part2)
    if lo:
        assert torch.equal(got["lo"], ref["lo"])
    P, R = part.view(F, nrec, 32, 3), pref.view(F, nrec, 32, 3)
    assert torch.equal(P[..., 0], R[..., 0])
    check("record mean", P[..., 1], R[..., 1], 2e-6, 2e-6)
    check("record M2", P[..., 2], R[..., 2], 1e-4 * float(R[..., 2].max()), 1e-5)
    gamma, beta = rnd(C, seed=4) * 0.5 + 1, rnd(C, seed=5) * 0.3
    y0, y1 = (torch.zeros(M, C, device=DEV, dtype=torch.float16) for _ in range(2))
    hip.groupnorm_apply(ref["o32"], C, F, Npix, C, ppc, pref, gamma, beta, 1e-5, 1, y0, C)
    hip.groupnorm_apply(ref["o32"], C, F, Npix, C, 128, part, gamma, beta, 1e-5, 1, y1, C, n_records=nrec)
    torch.cuda.synchronize()
    check("apply on the concat's records", y1, y0, 2e-3, 2e-3)
    e = outs()
    epart = torch.zeros(F * nrec * 96, device=DEV)
    emu.concat_add(a, C1, s_, c, C2, M, e["o32"], e["o16"], e["lo"], gn_part=epart, frames=F, ppc=ppc)
    check("records vs emu", P[..., 1], epart.view(F, nrec, 32, 3)[..., 1], 2e-6, 2e-6)


def test_rejects_cpu_tensors_and_bad_shapes():
    with pytest.raises(hip.PncError):
        hip.layernorm(torch.zeros(4, 64), 64, 4, 64, torch.ones(64), torch.zeros(64), 1e-5,
                      torch.zeros(4, 64, dtype=torch.float16), 64)
    a = rnd(64, 60, dtype=torch.float16)
    with pytest.raises(hip.PncError):     # K not a multiple of 8
        hip.gemm(a, a, M=64, N=64, K=60, lda=60, out32=torch.zeros(64, 64, device=DEV), ldc32=64)


def test_lo_planes_of_norms_and_helpers():
    """every producer of a precise operand: hi is bit-identical to the plain fp16 output, hi + lo * 2^-11 reproduces the
    fp32 value to ~2^-22 relative"""
    def rec(hi, lo):
        return hi.float() + lo.float() / 2048.0
    F, Npix, C, ppc = 2, 300, 320, 128
    x = rnd(F * Npix, C) * 1.7 + 0.9
    gamma, beta = rnd(C) * 0.5 + 1, rnd(C) * 0.3
    nchunk = (Npix + ppc - 1) // ppc
    part = torch.zeros(F * nchunk * 32 * 3, device=DEV)
    y0, y1, lo = (torch.zeros(F * Npix, C, device=DEV, dtype=torch.float16) for _ in range(3))
    hip.groupnorm_stats(x, C, F, Npix, C, ppc, part)
    hip.groupnorm_apply(x, C, F, Npix, C, ppc, part, gamma, beta, 1e-5, 1, y0, C)
    hip.groupnorm_apply(x, C, F, Npix, C, ppc, part, gamma, beta, 1e-5, 1, y1, C, lo)
    ref = torch.nn.functional.silu(torch.nn.functional.group_norm(x.view(F, Npix, C).permute(0, 2, 1).double(), 32,
                                                                  gamma.double(), beta.double(), 1e-5)).permute(0, 2, 1).reshape(-1, C)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    check("gn hi+lo", rec(y1, lo), ref.float(), 2e-5, 2e-5)
    B, T = 1, 2
    xt = rnd(B * T * Npix, C) * 1.3 - 0.4
    t0, t1, tlo = (torch.zeros(B * T * Npix, C, device=DEV, dtype=torch.float16) for _ in range(3))
    e1, elo = torch.zeros_like(t0), torch.zeros_like(t0)
    hip.groupnorm_temporal_silu(xt, B, T, Npix, C, gamma, beta, 1e-5, t0)
    hip.groupnorm_temporal_silu(xt, B, T, Npix, C, gamma, beta, 1e-5, t1, tlo)
    emu.groupnorm_temporal_silu(xt, B, T, Npix, C, gamma, beta, 1e-5, e1, elo)
    torch.cuda.synchronize()
    assert torch.equal(t0, t1)
    check("gnt hi+lo", rec(t1, tlo), rec(e1, elo), 2e-5, 2e-5)
    M = 513
    xl = rnd(M, C) * 2.1 + 0.5
    l0, l1, llo = (torch.zeros(M, C, device=DEV, dtype=torch.float16) for _ in range(3))
    hip.layernorm(xl, C, M, C, gamma, beta, 1e-5, l0, C)
    hip.layernorm(xl, C, M, C, gamma, beta, 1e-5, l1, C, llo)
    torch.cuda.synchronize()
    assert torch.equal(l0, l1)
    check("ln hi+lo", rec(l1, llo), torch.nn.functional.layer_norm(xl.double(), (C,), gamma.double(), beta.double(), 1e-5).float(),
          2e-5, 2e-5)
    # layout / elementwise helpers
    Fh, C1, C2, H, W, Cpad = 2, 4, 4, 8, 12, 8
    a, b, sc = rnd(Fh, C1, H, W) * 7, rnd(Fh, C2, H, W), rnd(Fh).abs() + 0.5
    oh, ol = (torch.zeros(Fh * H * W, Cpad, device=DEV, dtype=torch.float16) for _ in range(2))
    hip.nchw_to_tokens_f16(a, C1, b, C2, Fh, H * W, Cpad, oh, ol, a_scale=sc)
    torch.cuda.synchronize()
    full = torch.cat([a * sc.view(Fh, 1, 1, 1), b], 1).permute(0, 2, 3, 1).reshape(Fh * H * W, Cpad)
    assert torch.equal(oh, full.half())
    assert (rec(oh, ol) - full).abs().max().item() <= 2.0 ** -21 * full.abs().max().item()
    Mh, Ca, Cb = 300, 128, 64
    aa, ss, cc = rnd(Mh, Ca) * 5, rnd(Mh, Cb), rnd(Mh, Cb)
    o16, o16lo = (torch.zeros(Mh, Ca + Cb, device=DEV, dtype=torch.float16) for _ in range(2))
    hip.concat_add(aa, Ca, ss, cc, Cb, Mh, None, o16, o16lo)
    torch.cuda.synchronize()
    cat = torch.cat([aa, ss + cc], 1)
    assert torch.equal(o16, cat.half()) and (rec(o16, o16lo) - cat).abs().max().item() <= 2.0 ** -21 * cat.abs().max().item()
    y16, ylo = (torch.zeros(Mh, Ca, device=DEV, dtype=torch.float16) for _ in range(2))
    hip.cast_f16(aa, Mh * Ca, y16, ylo)
    torch.cuda.synchronize()
    assert torch.equal(y16, aa.half()) and (rec(y16, ylo) - aa).abs().max().item() <= 2.0 ** -21 * aa.abs().max().item()


@pytest.mark.parametrize("F,widths,K,silu_in,silu_out", [(16, [320, 320, 640, 1280, 1280, 320], 1280, False, False),
                                                         (5, [64, 128], 256, True, True), (21, [320] * 22 + [640] * 11, 1280, False, False)])
def test_linear_smallm_segments_is_the_single_launch_per_site(F, widths, K, silu_in, silu_out):
    """pnc_linear_smallm_segments (the emb_layers Linear of every ResBlock of a network in one launch): every site's [F, width] block
    bit-identical to pnc_linear_smallm on that site's weight rows, and equal to the emulation's fp32 product."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(F, K, generator=g).cuda()
    N = sum(widths)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    seg = [0]
    for wd in widths:
        seg.append(seg[-1] + wd)
    out = torch.full((N * F,), float("nan"), device="cuda")
    for m0 in range(0, F, 16):
        hip.linear_smallm_segments(a[m0:], K, w, bias, out, min(16, F - m0), m0, F, N, K, seg, silu_in, silu_out)
    ref_e = torch.empty(N * F)
    for m0 in range(0, F, 16):
        emu.linear_smallm_segments(a.cpu()[m0:], K, w.cpu(), bias.cpu(), ref_e, min(16, F - m0), m0, F, N, K, seg, silu_in, silu_out)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    for s0, s1 in zip(seg[:-1], seg[1:]):
        one = torch.empty(F, s1 - s0, device="cuda")
        for m0 in range(0, F, 16):
            hip.linear_smallm(a[m0:], K, w[s0:], bias[s0:], one[m0:], s1 - s0, min(16, F - m0), s1 - s0, K, silu_in, silu_out)
        torch.cuda.synchronize()
        blk = out[s0 * F:s1 * F].view(F, s1 - s0)
        assert torch.equal(blk, one)
        assert (blk.cpu() - ref_e[s0 * F:s1 * F].view(F, s1 - s0)).abs().max().item() < 2e-4
    # argument checks: a segment table that does not tile N, a start that is not a multiple of 4
    for bad in ([0, N - 4], [0, 6, N]):
        with pytest.raises(Exception):
            hip.linear_smallm_segments(a, K, w, bias, out, min(16, F), 0, F, N, K, bad, silu_in, silu_out)
