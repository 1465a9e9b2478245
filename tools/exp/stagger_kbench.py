"""Isolated A/B of PNC_OPT_GEMM_STAGGER on the library's own launches (round 5): each case is timed with the option at 0 and at 1,
interleaved, (a) "warm" — one operand set, launches back to back — and (b) "cold" — rotating over operand sets that exceed the
256 MB Infinity Cache, so the activations stream from HBM as they do inside the network."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import engine, hip  # noqa: E402

DEV = "cuda"


def h16(*shape, scale=0.5):
    return (torch.randn(*shape, device=DEV) * scale).half()


def timed(fns, iters=12, rounds=5):
    """fns: {label: callable(i)}; returns {label: median us}"""
    out = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, fn in fns.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            if r:
                out[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}


def case(name, make, flops, nset_cold):
    res = {}
    for mode, nset in (("warm", 1), ("cold", nset_cold)):
        sets = [make() for _ in range(nset)]

        def run(opt):
            def f(i):
                hip.set_option(hip.OPT_GEMM_STAGGER, opt)
                sets[i % nset]()
            return f
        t = timed({"0": run(0), "1": run(1)})
        hip.set_option(hip.OPT_GEMM_STAGGER, 8)
        res[mode] = t
        del sets
        torch.cuda.empty_cache()
    print(f"{name:44s} " + "  ".join(f"{m}: [0] {res[m]['0']:7.1f} us {flops / res[m]['0'] / 1e6:5.0f} TF  [1] {res[m]['1']:7.1f} us {flops / res[m]['1'] / 1e6:5.0f} TF"
                                      f" ({(res[m]['0'] / res[m]['1'] - 1) * 100:+.1f} %)" for m in res), flush=True)


def plain(M, N, K, res=False, o32=False, lo8=False, vt=False, geglu=False):
    def make():
        a = h16(M, K)
        w = h16(N, K, scale=K ** -0.5)
        kw = dict(M=M, N=N, K=K, lda=K)
        if lo8:
            alo = torch.zeros(M, K, device=DEV, dtype=torch.uint8)
            hip.cast_f16(a.float() * 1.0003, M * K, torch.zeros_like(a), alo)
            kw.update(a16_lo=alo, w_lo=engine.pk_lo8(w))
        if geglu:
            o = torch.zeros(M, N // 2, device=DEV, dtype=torch.float16)
            kw.update(out16=o, ldc16=N // 2, geglu=True, bias=torch.zeros(N, device=DEV))
        elif vt:
            C = N // 3
            qk, vtt = torch.zeros(M, 2 * C, device=DEV, dtype=torch.float16), torch.zeros(16, C, M // 16, device=DEV, dtype=torch.float16)
            kw.update(out16=qk, ldc16=2 * C, out16t=vtt, ldt=M // 16, t_rows=M // 16, t_gstride=C * (M // 16), n_split=2 * C)
        elif o32:
            x = torch.zeros(M, N, device=DEV)
            kw.update(out32=x, ldc32=N)
            if res:
                kw.update(res1=x, ldr1=N)
        else:
            o = torch.zeros(M, N, device=DEV, dtype=torch.float16)
            kw.update(out16=o, ldc16=N)
            if res:
                kw.update(res1=torch.zeros(M, N, device=DEV), ldr1=N)
        return lambda: hip.gemm(a, w, **kw)
    return make


def conv3(F, H, W, Cin, N, up=0, lo8=False):
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    M, K = F * Ho * Wo, 9 * Cin

    def make():
        x = h16(F, H, W, Cin)
        w = h16(N, K, scale=K ** -0.5)
        o = torch.zeros(M, N, device=DEV)
        kw = dict(M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=dict(Cin=Cin, Hin=H, Win=W, Hout=Ho, Wout=Wo, stride=1, upsample=up), out32=o, ldc32=N)
        if lo8:
            xlo = torch.zeros(F, H, W, Cin, device=DEV, dtype=torch.uint8)
            hip.cast_f16(x.float() * 1.0003, x.numel(), torch.zeros_like(x), xlo)
            kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))
        return lambda: hip.gemm(x, w, **kw)
    return make, 2.0 * M * N * K


def conv1(B, T, Npix, C, lo8=True):
    M, N, K = B * T * Npix, C, 3 * C

    def make():
        x = h16(M, C)
        w = h16(N, K, scale=K ** -0.5)
        o = torch.zeros(M, N, device=DEV)
        kw = dict(M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=T, Npix=Npix), res1=o, ldr1=N, out32=o, ldc32=N)
        if lo8:
            xlo = torch.zeros(M, C, device=DEV, dtype=torch.uint8)
            hip.cast_f16(x.float() * 1.0003, M * C, torch.zeros_like(x), xlo)
            kw.update(a16_lo=xlo, w_lo=engine.pk_lo8(w))
        return lambda: hip.gemm(x, w, **kw)
    return make, 2.0 * M * N * K


flt = sys.argv[1] if len(sys.argv) > 1 else ""
cases = [
    ("L2 ff2  12288x1280x5120 res o16", plain(12288, 1280, 5120, res=True), 2.0 * 12288 * 1280 * 5120, 4),
    ("L2 ff2  12288x1280x5120 o16 (no residual)", plain(12288, 1280, 5120), 2.0 * 12288 * 1280 * 5120, 4),
    ("L1 ff2  49152x640x2560 res o16 (tail split)", plain(49152, 640, 2560, res=True), 2.0 * 49152 * 640 * 2560, 3),
    ("L0 ff2  196608x320x1280 res o16", plain(196608, 320, 1280, res=True), 2.0 * 196608 * 320 * 1280, 2),
    ("L2 qkv  12288x3840x1280 V^T", plain(12288, 3840, 1280, vt=True), 2.0 * 12288 * 3840 * 1280, 5),
    ("L2 CxC  12288x1280x1280 res o32", plain(12288, 1280, 1280, res=True, o32=True), 2.0 * 12288 * 1280 * 1280, 6),
    ("L1 ff1  49152x5120x640 geglu (persistent)", plain(49152, 5120, 640, geglu=True), 2.0 * 49152 * 5120 * 640, 4),
    ("L3 ff2  3072x1280x5120 res o16", plain(3072, 1280, 5120, res=True), 2.0 * 3072 * 1280 * 5120, 6),
]
for name, mk, fl, ns in cases:
    if flt in name:
        case(name, mk, fl, ns)
for name, (mk, fl), ns in [("L0 conv3x3 196608x320x2880 (stencil tiles)", conv3(16, 32, 384, 320, 320), 3),
                           ("L1 conv3x3 49152x640x5760 (stencil tiles)", conv3(16, 16, 192, 640, 640), 4),
                           ("L2 conv3x3 12288x1280x11520 (stencil tiles)", conv3(16, 8, 96, 1280, 1280), 6),
                           ("L3 conv3x3 3072x1280x11520 (per tap, split K)", conv3(16, 4, 48, 1280, 1280), 6),
                           ("L0 upsample 196608x640x5760 lo8 (per tap)", conv3(16, 16, 192, 640, 640, up=1, lo8=True), 3),
                           ("L0 conv1d 196608x320x960 lo8", conv1(2, 8, 12288, 320), 2),
                           ("L2 conv1d 12288x1280x3840 lo8", conv1(2, 8, 768, 1280), 6)]:
    if flt in name:
        case(name, mk, fl, ns)
