"""Same-box interleaved A/B of the three forms of the GEGLU (FF1) GEMM: one tile per 8-wave workgroup (PNC_OPT_GEMM_PERSIST 0), one
persistent 8-wave workgroup per CU (1, round 3), two independent persistent 4-wave workgroups per CU (5, round 4 experiment), at the
three levels of BASELINE config 3.

    python tools/exp/ff1_2wg_ab.py [rounds]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = "cuda"


def main(rounds=3):
    print(torch.cuda.get_device_name(0))
    for li, (M, C) in enumerate([(196608, 320), (49152, 640), (12288, 1280)]):
        g = torch.Generator(device=DEV).manual_seed(li)
        a = torch.randn(M, C, device=DEV, generator=g).half()
        w = (torch.randn(8 * C, C, device=DEV, generator=g) * C ** -0.5).half()
        b = torch.randn(8 * C, device=DEV, generator=g)
        o = torch.zeros(M, 4 * C, device=DEV, dtype=torch.float16)
        fn = lambda: hip.gemm(a, w, M=M, N=8 * C, K=C, lda=C, bias=b, geglu=True, out16=o, ldc16=4 * C)   # noqa: E731
        for rd in range(rounds):
            for opt in (0, 1, 5):
                hip.set_option(hip.OPT_GEMM_PERSIST, opt)
                t = timeit(fn, iters=16, warm=3)
                print(f"L{li} FF1 M={M} N={8 * C} K={C} round {rd} persist={opt}  {t * 1e6:8.1f} us {2.0 * M * 8 * C * C / t / 1e12:7.1f} TFLOP/s", flush=True)
    hip.set_option(hip.OPT_GEMM_PERSIST, 3)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
