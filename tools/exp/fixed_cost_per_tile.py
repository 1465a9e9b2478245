"""Fixed cost per output tile of the level-0 GEMMs: time(K) = a + b K at fixed M, N.  a = prologue (first K tile's DMA latency) +
epilogue; b K = the main loop.  What a persistent kernel that prefetches the next tile's first K tile / overlaps the epilogue can win
is bounded by a."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
from tools.kbench import timeit, h16

M = 196608
for name, N, kind in (("ff1 geglu", 2560, "geglu"), ("proj res o32", 320, "res"), ("qkv o16", 960, "o16")):
    ts = []
    for K in (320, 640, 960, 1280):
        a, w = h16(M, K), h16(N, K)
        bias = torch.zeros(N, device="cuda")
        if kind == "geglu":
            o = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
            fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o, ldc16=N // 2)
        elif kind == "res":
            o = torch.zeros(M, N, device="cuda")
            fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, res1=o, ldr1=N, out32=o, ldc32=N)
        else:
            o = torch.empty(M, N, device="cuda", dtype=torch.float16)
            fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N)
        ts.append((K, timeit(fn, iters=16, warm=3) * 1e6))
        del a, w, o
    (k0, t0), (k1, t1) = ts[0], ts[-1]
    b = (t1 - t0) / (k1 - k0)
    print(f"{name:14s} M={M} N={N}: " + "  ".join(f"K={k}: {t:6.1f}us" for k, t in ts) + f"   -> a = {t0 - b * k0:6.1f} us, b = {b * 320:5.1f} us per 320 of K")
