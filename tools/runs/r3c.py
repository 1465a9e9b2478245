"""A/B of the 128x320 two-workgroups-per-CU geometry (PNC_OPT_GEMM_TILE 6) against 256x320 (3) on the stream-bound GEMM shapes of
BASELINE config 3 (profiles/round3/shape_profile_r3b_precise.log), interleaved per shape; plus the MALL panel experiment for FF."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from panacea_amd import engine, hip
DEV = "cuda"


def timeit(fn, iters=24, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e-3


def h16(*s):
    return (torch.randn(*s, device=DEV) * 0.5).half()


def case(name, M, N, K, mode=0, res=True, ln=False, lo8=False, o16=False, o32=True, res2=False, rb=False, T=8):
    a = h16(M, K if mode == 0 else K // 3)
    w = h16(N, K) * (K ** -0.5)
    kw = dict(a16=a, w16=w, M=M, N=N, K=K, bias=torch.zeros(N, device=DEV))
    if mode == 0:
        kw["lda"] = K
    else:
        kw.update(a_mode=hip.A_CONV1D_T, tconv=dict(C=K // 3, T=T, Npix=M // (2 * T)))
    o = torch.zeros(M, N, device=DEV)
    if o32:
        kw.update(out32=o, ldc32=N)
    if res:
        kw.update(res1=o, ldr1=N)
    if res2:
        kw.update(res2=torch.zeros(M, N, device=DEV), ldr2=N)
    if rb:
        kw.update(rowbias=torch.zeros(16, N, device=DEV), rb_rows=M // 16, rb_mod=16)
    if ln:
        kw.update(ln_gamma=torch.ones(N, device=DEV), ln_beta=torch.zeros(N, device=DEV),
                  ln_out16=torch.empty(M, N, device=DEV, dtype=torch.float16), ldln=N, ln_in_library=True)
    if lo8:
        kw.update(a16_lo=torch.randint(0, 255, a.shape, device=DEV, dtype=torch.uint8), w_lo=engine.pk_lo8(w))
    if o16:
        kw.update(out16=torch.empty(M, N, device=DEV, dtype=torch.float16), ldc16=N,
                  out16_lo=torch.empty(M, N, device=DEV, dtype=torch.uint8))
    ts = {}
    for rep in range(2):
        for tile in (3, 6):
            prev = hip.set_option(hip.OPT_GEMM_TILE, tile)
            t = timeit(lambda: hip.gemm(**kw), iters=16, warm=3)
            hip.set_option(hip.OPT_GEMM_TILE, prev)
            ts.setdefault(tile, []).append(t * 1e6)
    print(f"{name:44s} 256x320 {min(ts[3]):7.1f} us   128x320 x2/CU {min(ts[6]):7.1f} us   ({min(ts[6]) / min(ts[3]):.3f})", flush=True)


M0, M1, M2 = 196608, 49152, 12288
case("L0 attn out-proj res+ln        K=320", M0, 320, 320, ln=True)
case("L0 proj_in o32+ln lo8          K=320", M0, 320, 320, res=False, ln=True, lo8=True)
case("L0 proj_out res lo8            K=320", M0, 320, 320, lo8=True)
case("L0 ff2 res o16+lo8             K=1280", M0, 320, 1280, o16=True, o32=False)
case("L0 q-proj o16 (no fp32 stream) K=320", M0, 320, 320, res=False, o32=False, o16=True)
case("L0 conv1d res+rb lo8           K=960", M0, 320, 960, mode=2, rb=True, lo8=True)
case("L0 conv1d 2res lo8             K=960", M0, 320, 960, mode=2, res2=True, lo8=True)
case("L1 attn out-proj res (+ln lib) K=640", M1, 640, 640)
case("L1 proj_in o32 lo8             K=640", M1, 640, 640, res=False, lo8=True)
case("L1 ff2 res o16+lo8             K=2560", M1, 640, 2560, o16=True, o32=False)
case("L1 conv1d res+rb lo8           K=1920", M1, 640, 1920, mode=2, rb=True, lo8=True)
case("L2 attn out-proj res           K=1280", M2, 1280, 1280)
case("L2 ff2 res o16+lo8             K=5120", M2, 1280, 5120, o16=True, o32=False)
case("L2 conv1d res+rb lo8           K=3840", M2, 1280, 3840, mode=2, rb=True, lo8=True)
