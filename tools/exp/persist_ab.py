"""Same-box interleaved A/B of the persistent plain-A GEMM (PNC_OPT_GEMM_PERSIST bit 1, round 4) on the level-0 launches it serves
(M = 196 608 tokens, C = 320): C x C + residual + LayerNorm, proj_in (+ e4m3 lo pass) + norm1, proj_out + residual with lo pass,
FF2 (K = 4C) with fp16 + e4m3 outputs, q projection, QKV with channel-major V^T.

    python tools/exp/persist_ab.py [rounds]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import engine, hip  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda"


def cases(M, C):
    g = torch.Generator(device=DEV).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)          # noqa: E731
    a32, a4 = r(M, C), r(M, 4 * C).half()
    a = a32.half()
    alo = torch.zeros(M, C, device=DEV, dtype=torch.uint8)
    hip.cast_f16(a32, M * C, torch.zeros_like(a), alo)
    w, w4, w3 = (r(C, C) * C ** -0.5).half(), (r(C, 4 * C) * (4 * C) ** -0.5).half(), (r(3 * C, C) * C ** -0.5).half()
    wl = engine.pk_lo8(w)
    bias, gam, bet = r(C), r(C) * 0.3 + 1, r(C) * 0.3
    x = r(M, C)
    ln = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    o16, olo = torch.zeros(M, C, device=DEV, dtype=torch.float16), torch.zeros(M, C, device=DEV, dtype=torch.uint8)
    qk, vt = torch.zeros(M, 2 * C, device=DEV, dtype=torch.float16), torch.zeros(16, C, M // 16, device=DEV, dtype=torch.float16)
    lnkw = dict(ln_gamma=gam, ln_beta=bet, ln_out16=ln, ldln=C)
    yield "CxC res o32 ln", 2.0 * M * C * C, lambda: hip.gemm(a, w, M=M, N=C, K=C, lda=C, bias=bias, res1=x, ldr1=C, out32=x, ldc32=C, **lnkw)
    yield "CxC o32 lo8 ln (proj_in)", 2.0 * M * C * C, lambda: hip.gemm(a, w, M=M, N=C, K=C, lda=C, bias=bias, out32=x, ldc32=C, a16_lo=alo, w_lo=wl, **lnkw)
    yield "CxC res o32 lo8 (proj_out)", 2.0 * M * C * C, lambda: hip.gemm(a, w, M=M, N=C, K=C, lda=C, bias=bias, res1=x, ldr1=C, out32=x, ldc32=C, a16_lo=alo, w_lo=wl)
    yield "FF2 res o16+lo8 K=4C", 2.0 * M * C * 4 * C, lambda: hip.gemm(a4, w4, M=M, N=C, K=4 * C, lda=4 * C, bias=bias, res1=x, ldr1=C, out16=o16, ldc16=C, out16_lo=olo)
    yield "q o16", 2.0 * M * C * C, lambda: hip.gemm(a, w, M=M, N=C, K=C, lda=C, out16=o16, ldc16=C)
    yield "QKV T o16 N=3C", 2.0 * M * 3 * C * C, lambda: hip.gemm(a, w3, M=M, N=3 * C, K=C, lda=C, out16=qk, ldc16=2 * C, out16t=vt, ldt=M // 16,
                                                                 t_rows=M // 16, t_gstride=C * (M // 16), n_split=2 * C)


def main(rounds=3):
    print(torch.cuda.get_device_name(0))
    M, C = 196608, 320
    for name, flops, fn in cases(M, C):
        for rd in range(rounds):
            for opt in (1, 3):
                hip.set_option(hip.OPT_GEMM_PERSIST, opt)
                t = timeit(fn, iters=16, warm=3)
                print(f"{name:30s} round {rd} persist={opt}  {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TFLOP/s", flush=True)
    hip.set_option(hip.OPT_GEMM_PERSIST, 3)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
