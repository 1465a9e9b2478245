"""stencil_tile_kernel, fp32-only epilogue: straight from the accumulators (shipped since round 6) against staged through LDS
(PNC_OPT_GEMM_FUSE_LN + 2): bit identity, then interleaved timings on the 3x3 convs of config 3, rotated operand sets."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, timeit, h16  # noqa: E402

for li, cin_mul in ((0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (2, 1)):
    C, H, W = LEVELS[li]
    Cin = C * cin_mul
    M = F * H * W
    xs = [h16(F, H, W, Cin) for _ in range(3)]
    w = h16(C, 9 * Cin) * 0.05
    outs = [torch.empty(M, C, device="cuda") for _ in range(3)]
    conv = dict(Cin=Cin, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
    it = [0]

    def fn():
        i = it[0] % 3
        it[0] += 1
        hip.gemm(xs[i], w, M=M, N=C, K=9 * Cin, a_mode=hip.A_CONV3X3, conv=conv, out32=outs[i], ldc32=C)
    ref = {}
    for opt in (3, 1):
        prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, opt)
        o = torch.zeros(M, C, device="cuda")
        hip.gemm(xs[0], w, M=M, N=C, K=9 * Cin, a_mode=hip.A_CONV3X3, conv=conv, out32=o, ldc32=C)
        torch.cuda.synchronize()
        hip.set_option(hip.OPT_STENCIL_TILES, prev)
        ref[opt] = o
    ts = {3: [], 1: []}
    for rd in range(3):
        for opt in (3, 1):
            prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, opt)
            ts[opt].append(timeit(fn, iters=15, warm=3))
            hip.set_option(hip.OPT_STENCIL_TILES, prev)
    fl = 2.0 * M * C * 9 * Cin
    a, b = min(ts[3]), min(ts[1])
    print(f"L{li} conv3x3 {Cin}->{C} identical={torch.equal(ref[1], ref[3])}  staged-epilogue {a*1e6:7.1f} us {fl/a/1e12:6.0f} TF   direct-epilogue {b*1e6:7.1f} us {fl/b/1e12:6.0f} TF  {100*(b/a-1):+5.1f} %"
          f"   rounds {[f'{x*1e6:.0f}/{y*1e6:.0f}' for x, y in zip(ts[3], ts[1])]}", flush=True)
