"""(needs tools/exp/persist_hand_placed_reads.patch applied to panacea_amd/csrc/gemm_kernel.h)
gemm_persist_kernel with the fragment reads of k-step ks + 1 placed by hand into the MFMA stream of k-step ks (experiment,
PNC_OPT_GEMM_PERSIST bit 3) against hipcc's own schedule: bit identity, then interleaved timings, level-0 shapes (rotated operand sets)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, timeit  # noqa: E402

DEV = "cuda"
NBUF = 3


def h16(*shape, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, device=DEV, generator=g) * 0.5).half()


def case(li, name, M, N, K, kind):
    As = [h16(M, K, seed=i) for i in range(NBUF)]
    w = (h16(N, K, seed=9) * (K ** -0.5) * 2).contiguous()
    bias = torch.randn(N, device=DEV) * 0.1
    res = torch.randn(M, N, device=DEV)
    gamma, beta = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV) * 0.1

    def outs():
        o = {}
        if kind == "o16":
            o["h"] = torch.zeros(M, N, device=DEV, dtype=torch.float16)
        if kind == "res_o16":
            o["h"] = torch.zeros(M, N, device=DEV, dtype=torch.float16)
            o["lo"] = torch.zeros(M, N, device=DEV, dtype=torch.uint8)
        if kind == "res_ln":
            o["x"] = res.clone()
            o["ln"] = torch.zeros(M, N, device=DEV, dtype=torch.float16)
        return o

    def call(a, o):
        kw = dict(M=M, N=N, K=K, lda=K, bias=bias)
        if kind == "o16":
            kw.update(out16=o["h"], ldc16=N)
        if kind == "res_o16":
            kw.update(res1=res, ldr1=N, out16=o["h"], ldc16=N, out16_lo=o["lo"])
        if kind == "res_ln":
            kw.update(res1=o["x"], ldr1=N, out32=o["x"], ldc32=N, ln_gamma=gamma, ln_beta=beta, ln_out16=o["ln"], ldln=N)
        hip.gemm(a, w, **kw)

    # bit identity
    ref, got = outs(), outs()
    prev = hip.set_option(hip.OPT_GEMM_PERSIST, 3)
    call(As[0], ref)
    hip.set_option(hip.OPT_GEMM_PERSIST, 11)
    call(As[0], got)
    torch.cuda.synchronize()
    same = all(torch.equal(ref[k], got[k]) for k in ref)
    worst = max((ref[k].float() - got[k].float()).abs().max().item() for k in ref)
    # timings
    sets = [outs() for _ in range(NBUF)]
    it = [0]

    def fn():
        i = it[0] % NBUF
        it[0] += 1
        call(As[i], sets[i])
    res_t = {}
    for rd in range(3):
        for opt in (3, 11):
            hip.set_option(hip.OPT_GEMM_PERSIST, opt)
            res_t.setdefault(opt, []).append(timeit(fn, iters=18, warm=3))
    hip.set_option(hip.OPT_GEMM_PERSIST, prev)
    t3, t7 = min(res_t[3]), min(res_t[11])
    fl = 2.0 * M * N * K
    print(f"L{li} {name:8s} M={M} N={N} K={K} {kind:8s} identical={same} (max diff {worst:.2e})  two-stage {t3*1e6:7.1f} us {fl/t3/1e12:6.0f} TF"
          f"   hand-placed {t7*1e6:7.1f} us {fl/t7/1e12:6.0f} TF   {100*(t7/t3-1):+5.1f} %   rounds {[f'{a*1e6:.0f}/{b*1e6:.0f}' for a, b in zip(res_t[3], res_t[11])]}",
          flush=True)


for li, (C, H, W) in list(enumerate(LEVELS[:1])):
    M = F * H * W
    case(li, "ff2", M, C, 4 * C, "res_o16")
    case(li, "to_out", M, C, C, "res_ln")
    case(li, "q", M, C, C, "o16")
    case(li, "qkv", M, 3 * C, C, "o16")
