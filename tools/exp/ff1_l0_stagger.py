"""L0 / L1 FF1 (persistent GEGLU kernel) under PNC_OPT_GEMM_STAGGER values: is the K = 320 loop (5 K tiles) better staggered?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, timeit, h16  # noqa: E402
for li, (C, H, W) in enumerate(LEVELS[:2]):
    M = F * H * W
    a, w = h16(M, C), h16(8 * C, C)
    bias = torch.zeros(8 * C, device="cuda")
    o = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    fn = lambda: hip.gemm(a, w, M=M, N=8 * C, K=C, lda=C, bias=bias, geglu=True, out16=o, ldc16=4 * C)   # noqa: E731
    for rd in range(3):
        for st in (8, 0, 4, 1):
            prev = hip.set_option(hip.OPT_GEMM_STAGGER, st)
            t = timeit(fn, iters=16, warm=3)
            hip.set_option(hip.OPT_GEMM_STAGGER, prev)
            print(f"FF1 L{li} round {rd} stagger {st}: {t*1e6:7.1f} us {2.0*M*8*C*C/t/1e12:7.1f} TF", flush=True)
