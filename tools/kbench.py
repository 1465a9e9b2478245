"""Micro-benchmarks of the individual HIP kernels at the shapes of BASELINE config 3
(F = 16 panoramic frames; L0..L3).  Prints one line per case: time, TFLOP/s or GB/s.

    python tools/kbench.py [filter]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402

DEV = "cuda"
F = 16
LEVELS = [(320, 32, 384), (640, 16, 192), (1280, 8, 96), (1280, 4, 48)]


def timeit(fn, iters=24, warm=4):
    """median of per-launch HIP-event times (single-launch numbers on this box scatter by +-15 %)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e-3


def h16(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).half()


def report(name, t, flops=None, bytes_=None):
    s = f"{name:58s} {t * 1e6:10.1f} us"
    if flops:
        s += f"  {flops / t / 1e12:8.1f} TFLOP/s"
    if bytes_:
        s += f"  {bytes_ / t / 1e9:8.0f} GB/s"
    print(s, flush=True)



def bench_gemm(flt):
    for li, (C, H, W) in enumerate(LEVELS):
        M = F * H * W
        for name, N, K, kw in [("qkv", 3 * C, C, {}), ("proj", C, C, {"res": True}), ("ff1-geglu", 8 * C, C, {"geglu": True}),
                               ("ff2", C, 4 * C, {"res": True})]:
            tag = f"gemm L{li} {name} M={M} N={N} K={K}"
            if flt and flt not in tag:
                continue
            a, w = h16(M, K), h16(N, K)
            bias = torch.zeros(N, device=DEV)
            if kw.get("geglu"):
                o = torch.empty(M, N // 2, device=DEV, dtype=torch.float16)
                fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o, ldc16=N // 2)
            elif kw.get("res"):
                o = torch.zeros(M, N, device=DEV)
                fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, res1=o, ldr1=N, out32=o, ldc32=N)
            else:
                o = torch.empty(M, N, device=DEV, dtype=torch.float16)
                fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N)
            report(tag, timeit(fn), flops=2.0 * M * N * K)
            if kw.get("res") and name == "proj":
                # the same call-site with a precise (split) operand: K loop over the lo and the hi plane
                alo = h16(M, K)
                report(tag + " PRECISE", timeit(lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, res1=o, ldr1=N, out32=o,
                                                                 ldc32=N, a16_lo=alo)), flops=2.0 * M * N * K)
                del alo
            del a, w, o
        # conv3x3 C->C and temporal conv1d
        tag = f"conv3x3 L{li} C={C} {H}x{W}"
        if not flt or flt in tag:
            x, w = h16(F, H, W, C), h16(C, 9 * C)
            o = torch.empty(M, C, device=DEV)
            conv = dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
            report(tag, timeit(lambda: hip.gemm(x, w, M=M, N=C, K=9 * C, a_mode=hip.A_CONV3X3, conv=conv, out32=o, ldc32=C)),
                   flops=2.0 * M * C * 9 * C)
            xlo = h16(F, H, W, C)
            report(tag + " PRECISE", timeit(lambda: hip.gemm(x, w, M=M, N=C, K=9 * C, a_mode=hip.A_CONV3X3, conv=conv, out32=o,
                                                             ldc32=C, a16_lo=xlo)), flops=2.0 * M * C * 9 * C)
            del xlo
        tag = f"conv1d_t L{li} C={C}"
        if not flt or flt in tag:
            x, w = h16(M, C), h16(C, 3 * C)
            o = torch.zeros(M, C, device=DEV)
            report(tag, timeit(lambda: hip.gemm(x, w, M=M, N=C, K=3 * C, a_mode=hip.A_CONV1D_T,
                                                tconv=dict(C=C, T=8, Npix=H * W), res1=o, ldr1=C, out32=o, ldc32=C)),
                   flops=2.0 * M * C * 3 * C)


INTRA = [[0], [1], [2], [3], [4], [5]]
CROSS = [[5, 1], [0, 2], [1, 3], [2, 4], [3, 5], [4]]


def bench_attn(flt):
    for li, (C, H, W) in enumerate(LEVELS):
        N, heads, M = H * W, C // 64, F * H * W
        Nv = N // 6
        qk, vt, o = h16(M, 2 * C), h16(F, C, N), torch.empty(M, C, device=DEV, dtype=torch.float16)
        for name, segs, nk in [("intra", INTRA, 6 * Nv), ("cross", CROSS, 11 * Nv)]:
            tag = f"attn L{li} {name} Nv={Nv} heads={heads}"
            if flt and flt not in tag:
                continue
            fn = lambda: hip.attn_views(qk, 2 * C, qk[:, C:], 2 * C, vt, N, C * N, o, C, groups=F, heads=heads, H=H,
                                        W=W, views=6, kvH=H, kvW=W, kv_views=6, kv_rows_per_group=N, q_per_kv=1,
                                        kv_valid=Nv, segs=segs, scale=0.125)
            report(tag, timeit(fn), flops=4.0 * F * heads * Nv * nk * 64)
        tag = f"attn L{li} text heads={heads}"
        if not flt or flt in tag:
            kt, vtt = h16(2 * 80, C), h16(2, C, 80)
            fn = lambda: hip.attn_views(qk, 2 * C, kt, C, vtt, 80, C * 80, o, C, groups=F, heads=heads, H=H, W=W,
                                        views=1, kvH=1, kvW=80, kv_views=1, kv_rows_per_group=80, q_per_kv=8,
                                        kv_valid=77, segs=[[0]], scale=0.125)
            report(tag, timeit(fn), flops=4.0 * M * heads * 77 * 64)
        tag = f"attn L{li} temporal"
        if not flt or flt in tag:
            qkv = h16(M, 3 * C)
            fn = lambda: hip.attn_temporal(qkv, 3 * C, qkv[:, C:], 3 * C, qkv[:, 2 * C:], 3 * C, o, C, B=2, T=8,
                                           Npix=N, heads=heads, scale=0.125)
            report(tag, timeit(fn), bytes_=M * C * 2 * 4.0)


def bench_norm(flt):
    for li, (C, H, W) in enumerate(LEVELS):
        N, M = H * W, F * H * W
        x = torch.randn(M, C, device=DEV)
        y = torch.empty(M, C, device=DEV, dtype=torch.float16)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        ppc = max(16, min(128, N // 48))      # engine._ppc
        part = torch.empty(F * ((N + ppc - 1) // ppc) * 32 * 3, device=DEV)
        tag = f"groupnorm L{li} C={C}"
        if not flt or flt in tag:
            report(tag + " stats", timeit(lambda: hip.groupnorm_stats(x, C, F, N, C, ppc, part)), bytes_=M * C * 4.0)
            report(tag + " apply", timeit(lambda: hip.groupnorm_apply(x, C, F, N, C, ppc, part, g, b, 1e-5, 1, y, C)),
                   bytes_=M * C * 6.0)
        tag = f"gn_temporal L{li} C={C}"
        if not flt or flt in tag:
            report(tag, timeit(lambda: hip.groupnorm_temporal_silu(x, 2, 8, N, C, g, b, 1e-5, y)), bytes_=M * C * 6.0)
        tag = f"layernorm L{li} C={C}"
        if not flt or flt in tag:
            report(tag, timeit(lambda: hip.layernorm(x, C, M, C, g, b, 1e-5, y, C)), bytes_=M * C * 6.0)


def bench_tiles(flt):
    """every GEMM shape of the network's transformer blocks on every tile geometry (pnc_set_option PNC_OPT_GEMM_TILE):
    the table the score-based tile choice of gemm_kernel.h is tuned from"""
    names = {0: "auto", 1: "128x128", 2: "256x128", 3: "256x320", 4: "256x256"}
    for li, (C, H, W) in enumerate(LEVELS):
        M = F * H * W
        for name, N, K, kind in [("qkv", 3 * C, C, "vt"), ("q", C, C, "o16"), ("proj", C, C, "res"), ("ff1", 8 * C, C, "geglu"),
                                 ("ff2", C, 4 * C, "res"), ("conv1d", C, 3 * C, "conv1d"), ("conv3x3", C, 9 * C, "conv3x3")]:
            tag = f"tiles L{li} {name} M={M} N={N} K={K}"
            if flt and flt not in tag:
                continue
            w = h16(N, K)
            bias = torch.zeros(N, device=DEV)
            if kind == "conv3x3":
                a = h16(F, H, W, C)
            else:
                a = h16(M, K if kind != "conv1d" else C)
            o32 = torch.zeros(M, N if kind != "geglu" else 8, device=DEV) if kind in ("res", "conv1d", "conv3x3") else None
            o16 = torch.empty(M, (N // 2 if kind == "geglu" else (2 * C if kind == "vt" else N)), device=DEV, dtype=torch.float16) \
                if kind in ("vt", "o16", "geglu") else None
            vt = torch.empty(F, C, H * W, device=DEV, dtype=torch.float16) if kind == "vt" else None
            emb = torch.zeros(F, N, device=DEV)

            def fn():
                if kind == "vt":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o16, ldc16=2 * C, out16t=vt, ldt=H * W, t_rows=H * W,
                             t_gstride=C * H * W, n_split=2 * C)
                elif kind == "o16":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o16, ldc16=N)
                elif kind == "res":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, res1=o32, ldr1=N, out32=o32, ldc32=N)
                elif kind == "geglu":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o16, ldc16=N // 2)
                elif kind == "conv1d":
                    hip.gemm(a, w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=C, T=8, Npix=H * W), bias=bias,
                             rowbias=emb, rb_rows=H * W, rb_mod=F, res1=o32, ldr1=N, out32=o32, ldc32=N)
                else:
                    hip.gemm(a, w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, bias=bias, out32=o32, ldc32=N,
                             conv=dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0))
            res = []
            for t in (0, 1, 2, 3, 4):
                prev = hip.set_option(hip.OPT_GEMM_TILE, t)
                try:
                    res.append((names[t], timeit(fn, iters=12, warm=2)))
                finally:
                    hip.set_option(hip.OPT_GEMM_TILE, prev)
            fl = 2.0 * M * N * K
            print(f"{tag:48s} " + "  ".join(f"{n} {tt * 1e6:7.1f}us {fl / tt / 1e12:6.1f}TF" for n, tt in res), flush=True)
            del a, w


def bench_group_m(flt):
    """tile-order A/B (PNC_OPT_GEMM_GROUP_M): 1 = plain order, 0 = auto, 2 / 4 / 8 / 16 forced, on the shapes with many column tiles"""
    for li, (C, H, W) in enumerate(LEVELS):
        M = F * H * W
        for name, N, K, kind in [("ff1", 8 * C, C, "geglu"), ("qkv", 3 * C, C, "o16"), ("ff2", C, 4 * C, "res"),
                                 ("conv3x3", C, 9 * C, "conv3x3")]:
            tag = f"group_m L{li} {name} M={M} N={N} K={K}"
            if flt and flt not in tag:
                continue
            w = h16(N, K)
            bias = torch.zeros(N, device=DEV)
            a = h16(F, H, W, C) if kind == "conv3x3" else h16(M, K)
            o32 = torch.zeros(M, N, device=DEV) if kind in ("res", "conv3x3") else None
            o16 = torch.empty(M, N // 2 if kind == "geglu" else N, device=DEV, dtype=torch.float16) if kind in ("geglu", "o16") else None

            def fn():
                if kind == "geglu":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o16, ldc16=N // 2)
                elif kind == "o16":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o16, ldc16=N)
                elif kind == "res":
                    hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, res1=o32, ldr1=N, out32=o32, ldc32=N)
                else:
                    hip.gemm(a, w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, bias=bias, out32=o32, ldc32=N,
                             conv=dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0))
            res = []
            for g in (1, 0, 2, 4, 8, 16):
                prev = hip.set_option(hip.OPT_GEMM_GROUP_M, g)
                try:
                    res.append((g, timeit(fn, iters=12, warm=2)))
                finally:
                    hip.set_option(hip.OPT_GEMM_GROUP_M, prev)
            fl = 2.0 * M * N * K
            print(f"{tag:48s} " + "  ".join(f"g={g:2d} {tt * 1e6:7.1f}us {fl / tt / 1e12:6.1f}TF" for g, tt in res), flush=True)
            del a, w


def bench_halo(flt):
    """stencil-tile A/B (PNC_OPT_STENCIL_TILES): 0 = one gathered A tile per tap, 2 = tiles incl. halo, on the UNet / ControlNet
    conv3x3 shapes and the first-stage decoder's; the conv1d-T shapes (precise operand, as shipped) ride along as a reference"""
    shapes = []
    for li, (C, H, W) in enumerate(LEVELS):
        shapes.append((f"L{li} conv3x3 {C}->{C}", "c2", F, H, W, C, C))
        if li < 3:
            shapes.append((f"L{li} conv3x3 {2 * C}->{C}", "c2", F, H, W, 2 * C, C))
        shapes.append((f"L{li} conv1d {C} precise", "c1", F, H, W, C, C))
    shapes += [("vae 512->512 1/4", "c2", 2, 64, 768, 512, 512), ("vae 256->256 1/2", "c2", 2, 128, 1536, 256, 256),
               ("vae 128->128 1/1", "c2", 1, 256, 3072, 128, 128)]
    for name, kind, Fr, H, W, Cin, N in shapes:
        M, K = Fr * H * W, (9 if kind == "c2" else 3) * Cin
        tag = f"halo {name} M={M} N={N} K={K}"
        if flt and flt not in tag:
            continue
        a, w = h16(Fr, H, W, Cin), h16(N, K)
        alo = h16(Fr, H, W, Cin) if kind == "c1" else None
        bias = torch.zeros(N, device=DEV)
        o32 = torch.zeros(M, N, device=DEV)
        emb = torch.zeros(Fr, N, device=DEV)

        def fn():
            if kind == "c2":
                hip.gemm(a, w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, bias=bias, out32=o32, ldc32=N,
                         conv=dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0))
            else:
                hip.gemm(a, w, M=M, N=N, K=K, a_mode=hip.A_CONV1D_T, tconv=dict(C=Cin, T=8, Npix=H * W), bias=bias, a16_lo=alo,
                         rowbias=emb, rb_rows=H * W, rb_mod=Fr, res1=o32, ldr1=N, out32=o32, ldc32=N)
        res = []
        for opt in (0, 2, 0, 2):
            prev = hip.set_option(hip.OPT_STENCIL_TILES, opt)
            try:
                res.append((opt, timeit(fn, iters=10, warm=2)))
            finally:
                hip.set_option(hip.OPT_STENCIL_TILES, prev)
        fl = 2.0 * M * N * K
        print(f"{tag:52s} " + "  ".join(f"opt={o} {tt * 1e6:7.1f}us {fl / tt / 1e12:6.1f}TF" for o, tt in res), flush=True)
        del a, w, o32


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    print(torch.cuda.get_device_name(0))
    if flt.startswith("group_m"):
        bench_group_m(sys.argv[2] if len(sys.argv) > 2 else "")
        sys.exit(0)
    if flt.startswith("halo"):
        bench_halo(sys.argv[2] if len(sys.argv) > 2 else "")
        sys.exit(0)
    if flt.startswith("tiles"):
        bench_tiles(sys.argv[2] if len(sys.argv) > 2 else "")
        sys.exit(0)
    bench_gemm(flt)
    bench_attn(flt)
    bench_norm(flt)
