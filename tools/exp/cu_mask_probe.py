"""Round 6 probe: can two HIP streams with COMPLEMENTARY CU masks (hipExtStreamCreateWithCUMask) overlap an HBM-bound launch on
one half of the chip with an MFMA-bound launch on the other half — i.e. does a streaming kernel confined to 128 CUs pull more than
half of the HBM bandwidth when the other 128 CUs run a GEMM?  (Per-kernel overlap inside a CU is closed: 8 waves x 256 registers
and 144 KB of LDS fill it)

    python tools/exp/cu_mask_probe.py
"""
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402

DEV = "cuda"


def hip_runtime():
    """the libamdhip64 image this process already uses (torch's), not a second copy"""
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 not loaded")


def masked_stream(rt, words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def wall(fn_by_stream, reps=5):
    """fn_by_stream: [(stream, fn)] — every fn enqueued on its stream, wall time from a common start to all done (median)"""
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ends = []
        for st, fn in fn_by_stream:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                fn()
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ends.append(e)
        torch.cuda.synchronize()
        ts.append(max(e0.elapsed_time(e) for e in ends))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    print(torch.cuda.get_device_name(0))
    rt = hip_runtime()
    rt.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    rt.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    full = torch.cuda.Stream()
    full2 = torch.cuda.Stream()
    patterns = {
        "low128/high128": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
        "even/odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
        "nibbles": ([0x0F0F0F0F] * 8, [0xF0F0F0F0] * 8),
    }
    # operands: a streaming launch (LayerNorm, level 0) and an MFMA-bound launch (FF1 GEGLU, level 1), full and half batches
    C0, M0 = 320, 16 * 32 * 384
    x = torch.randn(M0, C0, device=DEV)
    y = torch.empty(M0, C0, device=DEV, dtype=torch.float16)
    g, b = torch.ones(C0, device=DEV), torch.zeros(C0, device=DEV)
    C1, M1 = 640, 16 * 16 * 192
    a = (torch.randn(M1, C1, device=DEV) * 0.5).half()
    w = (torch.randn(8 * C1, C1, device=DEV) * 0.5).half()
    bias = torch.zeros(8 * C1, device=DEV)
    o = torch.empty(M1, 4 * C1, device=DEV, dtype=torch.float16)
    # a residual GEMM of the stream class (level 0 proj + res, in place)
    ap = (torch.randn(M0, C0, device=DEV) * 0.5).half()
    wp = (torch.randn(C0, C0, device=DEV) * 0.5).half()
    bp = torch.zeros(C0, device=DEV)
    r32 = torch.zeros(M0, C0, device=DEV)

    def ln(rows, n):
        def f():
            for _ in range(n):
                hip.layernorm(x, C0, rows, C0, g, b, 1e-5, y, C0)
        return f

    def ff1(rows, n):
        def f():
            for _ in range(n):
                hip.gemm(a, w, M=rows, N=8 * C1, K=C1, lda=C1, bias=bias, geglu=True, out16=o, ldc16=4 * C1)
        return f

    def proj(rows, n):
        def f():
            for _ in range(n):
                hip.gemm(ap, wp, M=rows, N=C0, K=C0, lda=C0, bias=bp, res1=r32, ldr1=C0, out32=r32, ldc32=C0)
        return f

    for f in (ln(M0, 2), ff1(M1, 2), proj(M0, 2)):
        f()
    torch.cuda.synchronize()
    NL, NF, NP = 10, 1, 2          # ~ equal durations: 10 LayerNorms ~ 1 FF1 (level 1) ~ 2 proj (level 0)
    t_ln_full = wall([(full, ln(M0, NL))])
    t_ff_full = wall([(full, ff1(M1, NF))])
    t_pj_full = wall([(full, proj(M0, NP))])
    ln_bytes = M0 * C0 * 6.0 * NL
    print(f"full chip: {NL} x LN {t_ln_full:.3f} ms ({ln_bytes / t_ln_full / 1e9:.2f} TB/s)   {NF} x FF1 {t_ff_full:.3f} ms   {NP} x proj {t_pj_full:.3f} ms")
    t_free = wall([(full, ln(M0, NL)), (full2, ff1(M1, NF))])
    print(f"two unmasked streams, LN || FF1: {t_free:.3f} ms (sum of the two alone {t_ln_full + t_ff_full:.3f})")
    for name, (ma, mb) in patterns.items():
        try:
            sa, sb = masked_stream(rt, ma), masked_stream(rt, mb)
        except Exception as e:       # noqa: BLE001
            print(name, "FAILED:", e)
            continue
        t_ln_half = wall([(sa, ln(M0 // 2, NL))])
        t_ln_halfchip_full = wall([(sa, ln(M0, NL))])
        t_ff_half = wall([(sb, ff1(M1 // 2, NF))])
        t_pj_half = wall([(sa, proj(M0 // 2, NP))])
        # the experiment: half-batch streaming on A || half-batch GEMM on B, twice each (= the full batch's work)
        t_mix = wall([(sa, lambda: (ln(M0 // 2, NL)(), ff1(M1 // 2, NF)())), (sb, lambda: (ff1(M1 // 2, NF)(), ln(M0 // 2, NL)()))])
        t_same = wall([(sa, lambda: (ln(M0 // 2, NL)(), ff1(M1 // 2, NF)())), (sb, lambda: (ln(M0 // 2, NL)(), ff1(M1 // 2, NF)()))])
        t_mix_p = wall([(sa, lambda: (proj(M0 // 2, NP)(), ff1(M1 // 2, NF)())), (sb, lambda: (ff1(M1 // 2, NF)(), proj(M0 // 2, NP)()))])
        t_same_p = wall([(sa, lambda: (proj(M0 // 2, NP)(), ff1(M1 // 2, NF)())), (sb, lambda: (proj(M0 // 2, NP)(), ff1(M1 // 2, NF)()))])
        print(f"[{name}] 128 CUs alone: LN half batch {t_ln_half:.3f} ms ({ln_bytes / 2 / t_ln_half / 1e9:.2f} TB/s), LN full batch "
              f"{t_ln_halfchip_full:.3f} ms ({ln_bytes / t_ln_halfchip_full / 1e9:.2f} TB/s), FF1 half {t_ff_half:.3f} ms, proj half {t_pj_half:.3f} ms")
        print(f"[{name}] LN+FF1 both halves: anti-phased {t_mix:.3f} ms, in phase {t_same:.3f} ms, full-chip sequential {t_ln_full + t_ff_full:.3f} ms")
        print(f"[{name}] proj+FF1 both halves: anti-phased {t_mix_p:.3f} ms, in phase {t_same_p:.3f} ms, full-chip sequential {t_pj_full + t_ff_full:.3f} ms")


if __name__ == "__main__":
    main()
