"""time(K) of the library's plain fp16-output GEMM with PNC_OPT_GEMM_STAGGER 0 / 1 (round 5): slope = marginal main-loop rate, intercept =
per-launch fixed cost — which of the two the staggered schedule changes."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip  # noqa: E402
from tools.exp.stagger_kbench import timed, h16  # noqa: E402

DEV = "cuda"
for M, N, tile in ((196608, 320, 3), (49152, 1280, 3), (12288, 1280, 4), (12288, 1280, 3), (65536, 1280, 4)):
    hip.set_option(hip.OPT_GEMM_TILE, tile)
    hip.set_option(hip.OPT_GEMM_PERSIST, 0)
    rows = []
    for K in (640, 1280, 2560, 5120):
        a, w = h16(M, K), h16(N, K, scale=K ** -0.5)
        o = torch.zeros(M, N, device=DEV, dtype=torch.float16)

        def run(opt):
            def f(i):
                hip.set_option(hip.OPT_GEMM_STAGGER, opt)
                hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N)
            return f
        t = timed({"0": run(0), "1": run(1)}, iters=10, rounds=5)
        rows.append((K, t["0"], t["1"]))
        del a, w, o
    hip.set_option(hip.OPT_GEMM_STAGGER, 8)
    (k0, a0, b0), (k1, a1, b1) = rows[1], rows[-1]
    fl = 2.0 * M * N
    print(f"M={M} N={N} tile {'256x320' if tile == 3 else '256x256'}: " + "  ".join(f"K={k}: {x:.1f} / {y:.1f} us" for k, x, y in rows))
    print(f"    marginal rate K {k0}->{k1}: classic {fl * (k1 - k0) / (a1 - a0) / 1e6:.0f} TF, staggered {fl * (k1 - k0) / (b1 - b0) / 1e6:.0f} TF; "
          f"intercept (K -> 0) classic {a0 - (a1 - a0) / (k1 - k0) * k0:.1f} us, staggered {b0 - (b1 - b0) / (k1 - k0) * k0:.1f} us", flush=True)
hip.set_option(hip.OPT_GEMM_TILE, 0)
hip.set_option(hip.OPT_GEMM_PERSIST, 3)
