"""PNC_OPT_GEMM_PERSIST on / off on the FF1 shapes of config 3, interleaved."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
from tools.kbench import timeit, h16

for name, M, N, K in (("L0 ff1", 196608, 2560, 320), ("L1 ff1", 49152, 5120, 640), ("L2 ff1", 12288, 10240, 1280)):
    a, w = h16(M, K), h16(N, K)
    bias = torch.zeros(N, device="cuda")
    o = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
    fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o, ldc16=N // 2)
    res = []
    for v in (0, 1, 0, 1):
        prev = hip.set_option(hip.OPT_GEMM_PERSIST, v)
        try:
            res.append((v, timeit(fn, iters=20, warm=3) * 1e6))
        finally:
            hip.set_option(hip.OPT_GEMM_PERSIST, prev)
    print(f"{name} M={M} N={N} K={K}: " + "  ".join(f"persist={v}: {t:6.1f} us" for v, t in res), flush=True)
