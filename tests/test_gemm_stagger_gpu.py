"""PNC_OPT_GEMM_STAGGER (round 5): the staggered schedule of the 8-wave two-stage GEMM kernels — four phases per K tile, waves 4-7 one
barrier behind waves 0-3 — keeps the K order and the MFMA order of every accumulator, so every launch must equal the round-4
loops BIT FOR BIT: plain A on both 8-wave tiles (incl. tail-split rounds, K tails, the generic epilogue, split K), both precise
operand formats (e4m3 lo tiles phased per 64-k window; fp16 lo tiles + the 2^-11 rescale between the passes), the per-tap conv3x3
gathers (stride 2, nearest x2, e4m3 lo pass), the temporal conv (e4m3 lo pass, GroupNorm records from its epilogue) and the GEGLU
GEMM in its persistent and one-tile-per-workgroup forms.  Each case also runs twice staggered (no state left behind, no race that
shows as an occasional tile) and against the emulation."""
import pytest
import torch

import emu
from panacea_amd import engine, hip
from test_kernels_gpu import DEV, check, rnd

pytestmark = pytest.mark.gpu


def _ab(run, reps=3):
    """run() under OPT_GEMM_STAGGER = 0 (round 4's loops) and = 1 (staggered whenever the kernel has the schedule)"""
    prev = hip.set_option(hip.OPT_GEMM_STAGGER, 0)
    try:
        ref = run()
        hip.set_option(hip.OPT_GEMM_STAGGER, 1)
        got = [run() for _ in range(reps)]
        hip.set_option(hip.OPT_GEMM_STAGGER, 257)      # + 256 (round 6): the persistent GEGLU kernel's epilogue through its LDS slab
        got.append(run())
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_GEMM_STAGGER, prev)
    for g in got:
        for k in ref:
            assert torch.equal(g[k], ref[k]), f"{k}: staggered schedule differs from the round-4 loop"
    return ref


@pytest.mark.parametrize("M,N,K,tile", [(49152, 640, 640, 0), (1024, 640, 1280, 3), (12288, 1280, 2560, 0), (2048, 1280, 1280, 4),
                                        (700, 960, 608, 3), (4096, 320, 320, 3), (256, 1280, 5120, 0)])
@pytest.mark.parametrize("lo", [None, "e4m3", "f16"])
def test_plain_gemm_staggered_is_bit_identical(M, N, K, tile, lo):
    if lo == "e4m3" and K % 16:
        pytest.skip("e4m3 planes: 16-byte rows")
    a32 = rnd(M, K, seed=1)
    a = a32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    kw = dict(M=M, N=N, K=K, lda=K, bias=bias)
    if lo == "e4m3":
        alo = torch.zeros(M, K, device=DEV, dtype=torch.uint8)
        hip.cast_f16(a32, M * K, torch.zeros_like(a), alo)
        kw.update(a16_lo=alo, w_lo=engine.pk_lo8(w))
    elif lo == "f16":
        alo = torch.zeros(M, K, device=DEV, dtype=torch.float16)
        hip.cast_f16(a32, M * K, torch.zeros_like(a), alo)
        kw.update(a16_lo=alo)
    ptile = hip.set_option(hip.OPT_GEMM_TILE, tile)
    ppers = hip.set_option(hip.OPT_GEMM_PERSIST, 0)

    def run():
        o32, o16 = res.clone(), torch.zeros(M, N, device=DEV, dtype=torch.float16)
        hip.gemm(a, w, res1=o32, ldr1=N, out32=o32, ldc32=N, out16=o16, ldc16=N, **kw)
        return dict(o32=o32, o16=o16)
    try:
        ref = _ab(run)
    finally:
        hip.set_option(hip.OPT_GEMM_TILE, ptile)
        hip.set_option(hip.OPT_GEMM_PERSIST, ppers)
    if lo is None:
        e32, e16 = res.clone(), torch.zeros(M, N, device=DEV, dtype=torch.float16)
        emu.gemm(a, w, res1=e32, ldr1=N, out32=e32, ldc32=N, out16=e16, ldc16=N, **kw)
        check("vs emu", ref["o32"], e32, 3e-3)


def test_generic_epilogue_and_splitk_staggered_are_bit_identical():
    # ragged everything: the scalar epilogue variant on the 256 x 320 tile is never chosen for small N; force a wide ragged N
    M, N, K = 1000, 636, 1288
    a = rnd(M, K, dtype=torch.float16, seed=5)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=6)
    bias = rnd(N, seed=7)

    def run():
        o32 = torch.zeros(M, N, device=DEV)
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, out32=o32, ldc32=N)
        return dict(o32=o32)
    _ab(run)
    # split K on the 256 x 256 tile: each K slice runs its own staggered loop
    M, N, K = 256, 1280, 5120
    a = rnd(M, K, dtype=torch.float16, seed=8)
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=9)

    def run2():
        o32 = torch.zeros(M, N, device=DEV)
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, out32=o32, ldc32=N)          # (hip.gemm allocates the split-K workspace the library asks for)
        return dict(o32=o32)
    _ab(run2)


@pytest.mark.parametrize("F,Hin,Win,Cin,N,stride,up,lo8", [
    (4, 32, 96, 320, 320, 1, 0, False),        # per-tap gather of a stride-1 conv (stencil tiles switched off)
    (2, 32, 192, 320, 320, 2, 0, True),        # Downsample of a precise stream
    (2, 16, 96, 640, 640, 1, 1, True),         # Upsample: nearest x2 + conv, e4m3 lo pass
    (2, 16, 48, 128, 128, 1, 0, False),        # 128 x 128 tile: no staggered schedule there, must simply not change
])
def test_conv3x3_gather_is_option_independent(F, Hin, Win, Cin, N, stride, up, lo8):
    """(see test_conv1d_is_option_independent: the gather modes have one K loop; the option must not reach them — and they agree
    with the emulation)"""
    if up:
        Hout, Wout = 2 * Hin, 2 * Win
    else:
        Hout, Wout = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
    M, K = F * Hout * Wout, 9 * Cin
    x32 = rnd(F, Hin, Win, Cin, seed=11)
    x = x32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=12)
    bias = rnd(N, seed=13)
    conv = dict(Cin=Cin, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=stride, upsample=up)
    kw = dict(M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias)
    if lo8:
        xlo = torch.zeros(F, Hin, Win, Cin, device=DEV, dtype=torch.uint8)
        hip.cast_f16(x32, x32.numel(), torch.zeros_like(x), xlo)
        kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))
    pst = hip.set_option(hip.OPT_STENCIL_TILES, 0)

    def run():
        o = torch.zeros(M, N, device=DEV)
        hip.gemm(x, w, out32=o, ldc32=N, **kw)
        return dict(o=o)
    try:
        ref = _ab(run)
    finally:
        hip.set_option(hip.OPT_STENCIL_TILES, pst)
    if not lo8:
        e = torch.zeros(M, N, device=DEV)
        emu.gemm(x, w, out32=e, ldc32=N, **kw)
        check("conv3x3 vs emu", ref["o"], e, 3e-3)


@pytest.mark.parametrize("B,T,Npix,C,lo8,gs", [(2, 8, 3072, 320, True, True), (1, 8, 768, 640, True, False), (2, 4, 1024, 320, False, False)])
def test_conv1d_is_option_independent(B, T, Npix, C, lo8, gs):
    """(Round 5 compared two schedules here; since the staggered loop was confined to PNC_A_PLAIN the gather modes run ONE loop whatever
    PNC_OPT_GEMM_STAGGER says — ADVICE r5: what this still pins is that the option does not reach them: same bits for every value.)"""
    M, N, K = B * T * Npix, C, 3 * C
    x32 = rnd(M, C, seed=21)
    x = x32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=22)
    bias, emb, res = rnd(N, seed=23), rnd(B * T, N, seed=24), rnd(M, N, seed=25)
    kw = dict(M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), bias=bias, rowbias=emb, rb_rows=Npix, rb_mod=B * T)
    if lo8:
        xlo = torch.zeros(M, C, device=DEV, dtype=torch.uint8)
        hip.cast_f16(x32, M * C, torch.zeros_like(x), xlo)
        kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))

    def run():
        o = res.clone()
        out = dict(o=o)
        k2 = dict(kw)
        if gs:
            out["part"] = torch.zeros(B * T * (Npix // 64) * 32 * 3, device=DEV)
            k2.update(gn_part=out["part"])
        hip.gemm(x, w, res1=o, ldr1=N, out32=o, ldc32=N, **k2)
        return out
    _ab(run)


@pytest.mark.parametrize("M,C", [(49152, 640), (3072, 1280), (8192, 320)])
def test_geglu_staggered_is_bit_identical(M, C):
    """M = 49152: >= 512 tiles -> the persistent kernel (K loop staggered per output tile, groups re-aligned before the epilogue,
    next tile's first K tile requested in the last K tile's phases); the others: one tile per workgroup"""
    a = rnd(M, C, dtype=torch.float16, seed=31)
    w = rnd(8 * C, C, scale=C ** -0.5, dtype=torch.float16, seed=32)     # (value / gate row blocks as they lie: any weights do)
    bias = rnd(8 * C, seed=33)

    def run():
        o = torch.zeros(M, 4 * C, device=DEV, dtype=torch.float16)
        hip.gemm(a, w, M=M, N=8 * C, K=C, lda=C, bias=bias, out16=o, ldc16=4 * C, geglu=True)
        return dict(o=o)
    _ab(run)


@pytest.mark.parametrize("F,H,W,Cin,N,lo", [(2, 32, 64, 320, 320, None), (3, 8, 64, 128, 256, None), (2, 16, 16, 128, 320, "f16"),
                                            (12, 64, 96, 64, 320, None), (5, 32, 64, 64, 640, None), (16, 16, 192, 640, 640, None)])
def test_stencil_tile_kernel_staggered_is_bit_identical(F, H, W, Cin, N, lo):
    """gemm_stencil_tile.hip: one phase per k-step with the group offset vs the round-2 software pipeline (one barrier per half tile) —
    incl. tail-split rounds (quarter / half tile workgroups: one group idles), the 8 x 32 tile, 256-column tiles and a precise operand"""
    M, K = F * H * W, 9 * Cin
    x32 = rnd(F, H, W, Cin, seed=41)
    x = x32.half()
    w = rnd(N, K, scale=K ** -0.5, dtype=torch.float16, seed=42)
    bias, res = rnd(N, seed=43), rnd(M, N, seed=44)
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
    kw = dict(M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias)
    if lo:
        xlo = torch.zeros(F, H, W, Cin, device=DEV, dtype=torch.float16)
        hip.cast_f16(x32, x32.numel(), torch.zeros_like(x), xlo)
        kw.update(a16_lo=xlo)
    pst = hip.set_option(hip.OPT_STENCIL_TILES, 2)

    def run():
        o = res.clone()
        o16 = torch.zeros(M, N, device=DEV, dtype=torch.float16)
        hip.gemm(x, w, res1=o, ldr1=N, out32=o, ldc32=N, out16=o16, ldc16=N, **kw)
        return dict(o=o, o16=o16)
    pstag = hip.set_option(hip.OPT_GEMM_STAGGER, hip.set_option(hip.OPT_GEMM_STAGGER, 0))      # read the current value, leave it as it is
    try:
        ref = _ab(run)
        hip.set_option(hip.OPT_STENCIL_TILES, 0)
        hip.set_option(hip.OPT_GEMM_STAGGER, 0)
        tap = run()
        torch.cuda.synchronize()
    finally:
        hip.set_option(hip.OPT_STENCIL_TILES, pst)
        hip.set_option(hip.OPT_GEMM_STAGGER, pstag)
    assert torch.equal(tap["o"], ref["o"])            # and both equal the per-tap gather
