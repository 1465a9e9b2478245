"""What is FF1's per-tile fixed cost made of?  Same M, same 256x256 tiles: GEGLU epilogue (N = 2560 -> 1280 fp16 columns out) against a
plain fp16-output epilogue with the same number of tiles (N = 2560 out) and with the same output bytes (N = 1280)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
from tools.kbench import timeit, h16

M = 196608
prev = hip.set_option(hip.OPT_GEMM_TILE, 4)          # 256x256 for every case (what GEGLU runs on)
try:
    for name, N, geglu in (("geglu N=2560", 2560, True), ("o16 N=2560", 2560, False), ("o16 N=1280", 1280, False)):
        ts = []
        for K in (320, 1280):
            a, w = h16(M, K), h16(N, K)
            bias = torch.zeros(N, device="cuda")
            No = N // 2 if geglu else N
            o = torch.empty(M, No, device="cuda", dtype=torch.float16)
            fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=geglu, out16=o, ldc16=No)
            ts.append((K, timeit(fn, iters=16, warm=3) * 1e6))
            del a, w, o
        (k0, t0), (k1, t1) = ts
        b = (t1 - t0) / (k1 - k0)
        tiles = (M // 256) * ((N + 255) // 256)
        print(f"{name:14s}: K=320 {t0:6.1f} us, K=1280 {t1:6.1f} us -> a = {t0 - b * k0:6.1f} us = {(t0 - b * k0) / (tiles / 256):5.2f} us per tile round ({tiles} tiles), "
              f"b = {b * 320:5.1f} us per 320 of K")
finally:
    hip.set_option(hip.OPT_GEMM_TILE, prev)
