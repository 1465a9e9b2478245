"""Plain-A GEMMs with fp32 output (+ residual in place): the DIRECT epilogue (acc + bias + residual straight from the accumulators through
buffer loads / stores; shipped) against the one staged through LDS (PNC_OPT_GEMM_FUSE_LN + 2): bit identity, then interleaved timings at the
level-1 / level-2 shapes that run on gemm_glds_kernel (rotated operand sets)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, timeit  # noqa: E402

DEV = "cuda"


def h16(*shape, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * 0.5).half()


def case(li, name, M, N, K, res):
    As = [h16(M, K, seed=i) for i in range(3)]
    w = (h16(N, K, seed=9) * (K ** -0.5) * 2).contiguous()
    bias = torch.randn(N, device=DEV) * 0.1
    res0 = torch.randn(M, N, device=DEV)

    def call(a, o):
        kw = dict(M=M, N=N, K=K, lda=K, bias=bias, out32=o, ldc32=N)
        if res:
            kw.update(res1=o, ldr1=N)
        hip.gemm(a, w, **kw)
    outs = {}
    for opt in (3, 1):
        prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, opt)
        o = res0.clone()
        call(As[0], o)
        torch.cuda.synchronize()
        hip.set_option(hip.OPT_GEMM_FUSE_LN, prev)
        outs[opt] = o
    same = torch.equal(outs[1], outs[3])
    sets = [res0.clone() for _ in range(3)]
    it = [0]

    def fn():
        i = it[0] % 3
        it[0] += 1
        call(As[i], sets[i])
    ts = {3: [], 1: []}
    for rd in range(3):
        for opt in (3, 1):
            prev = hip.set_option(hip.OPT_GEMM_FUSE_LN, opt)
            ts[opt].append(timeit(fn, iters=18, warm=3))
            hip.set_option(hip.OPT_GEMM_FUSE_LN, prev)
    a, b = min(ts[3]), min(ts[1])
    print(f"L{li} {name:10s} M={M} N={N} K={K} res={res} identical={same} (max diff {(outs[1]-outs[3]).abs().max().item():.1e})  staged {a*1e6:7.1f} us   direct {b*1e6:7.1f} us"
          f"  {100*(b/a-1):+5.1f} %   rounds {[f'{x*1e6:.0f}/{y*1e6:.0f}' for x, y in zip(ts[3], ts[1])]}", flush=True)


for li, (C, H, W) in list(enumerate(LEVELS[:3])):
    M = F * H * W
    case(li, "to_out", M, C, C, True)
    case(li, "ff2-f32", M, C, 4 * C, True)
    case(li, "proj", M, C, C, False)
