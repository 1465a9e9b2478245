"""(needs tools/exp/ff1_phase_length.patch applied to panacea_amd/csrc/gemm_kernel.h)
FF1 (persistent GEGLU kernel), staggered K loop with ONE k-step per phase (shipped) against TWO k-steps per phase (PNC_OPT_GEMM_STAGGER
+ 256: half the barriers, 16 MFMAs per phase): bit identity, then interleaved timings at levels 0-2."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, timeit, h16  # noqa: E402
for li, (C, H, W) in enumerate(LEVELS[:3]):
    M = F * H * W
    As = [h16(M, C) for _ in range(3)]
    w = h16(8 * C, C) * 0.1
    bias = torch.randn(8 * C, device="cuda") * 0.1
    outs = [torch.empty(M, 4 * C, device="cuda", dtype=torch.float16) for _ in range(3)]
    it = [0]

    def fn():
        i = it[0] % 3
        it[0] += 1
        hip.gemm(As[i], w, M=M, N=8 * C, K=C, lda=C, bias=bias, geglu=True, out16=outs[i], ldc16=4 * C)
    ref = {}
    for st in (4, 516, 260):
        prev = hip.set_option(hip.OPT_GEMM_STAGGER, st)
        o = torch.zeros(M, 4 * C, device="cuda", dtype=torch.float16)
        hip.gemm(As[0], w, M=M, N=8 * C, K=C, lda=C, bias=bias, geglu=True, out16=o, ldc16=4 * C)
        torch.cuda.synchronize()
        hip.set_option(hip.OPT_GEMM_STAGGER, prev)
        ref[st] = o
    print(f"L{li}: identical 4 vs 260: {torch.equal(ref[4], ref[260])}, 4 vs 516: {torch.equal(ref[4], ref[516])}", flush=True)
    for rd in range(3):
        for st in (4, 516, 260):
            prev = hip.set_option(hip.OPT_GEMM_STAGGER, st)
            t = timeit(fn, iters=18, warm=3)
            hip.set_option(hip.OPT_GEMM_STAGGER, prev)
            print(f"FF1 L{li} round {rd} stagger {st:3d}: {t*1e6:7.1f} us {2.0*M*8*C*C/t/1e12:7.1f} TF", flush=True)
