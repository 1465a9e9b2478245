"""(round-3 record: needs the library of commit 07bbb9c, which still exported pnc_ff_chain_f16)
fused feed-forward launch (pnc_ff_chain_f16) vs LayerNorm + GEGLU GEMM + output GEMM at the level-0 shape of BASELINE config 3"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from panacea_amd import engine, hip
DEV = "cuda"


def timeit(fn, iters=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3


C, inner = 320, 1280
for M in (196608, 49152):
    x = torch.randn(M, C, device=DEV)
    w1 = (torch.randn(2 * inner, C, device=DEV) * C ** -0.5).half().float()
    w2 = (torch.randn(C, inner, device=DEV) * inner ** -0.5).half().float()
    b1, b2 = torch.zeros(2 * inner, device=DEV), torch.zeros(C, device=DEV)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    tape = engine.pk_ff_chain(w1, w2)
    w1i, b1i = engine.pk_geglu(w1, b1)
    w2h = w2.half()
    a16 = torch.empty(M, C, device=DEV, dtype=torch.float16)
    hid = torch.empty(M, inner, device=DEV, dtype=torch.float16)
    o16 = torch.empty(M, C, device=DEV, dtype=torch.float16)
    lo8 = torch.empty(M, C, device=DEV, dtype=torch.uint8)
    o32 = torch.empty(M, C, device=DEV)

    def fused_last():
        hip.ff_chain(x, C, M, C, inner, g, b, 1e-5, tape, b1, b2, out16=o16, ldo16=C, out16_lo=lo8)

    def fused_o32():
        hip.ff_chain(x, C, M, C, inner, g, b, 1e-5, tape, b1, b2, out32=o32, ldo32=C)

    def unfused_last():
        hip.layernorm(x, C, M, C, g, b, 1e-5, a16, C)
        hip.gemm(a16, w1i, M=M, N=2 * inner, K=C, lda=C, bias=b1i, geglu=True, out16=hid, ldc16=inner)
        hip.gemm(hid, w2h, M=M, N=C, K=inner, lda=inner, bias=b2, res1=x, ldr1=C, out16=o16, ldc16=C, out16_lo=lo8)

    fl = 2.0 * M * C * inner * 3
    for name, fn in (("fused, fp16 + e4m3 out (the network's case)", fused_last), ("fused, fp32 out", fused_o32),
                     ("LayerNorm + GEGLU GEMM + output GEMM", unfused_last)):
        t = timeit(fn)
        print(f"M={M:7d} {name:48s} {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
