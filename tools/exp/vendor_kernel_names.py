"""Which Tensile kernels the vendor GEMM (torch.mm -> hipBLASLt) picks for the three main-loop-bound shapes of VERDICT r4 item 1, plus
4096^3: run under `rocprofv3 --kernel-trace`; the kernel NAMES encode macro tile (MT), MFMA shape (MI), wave tile (MIWT), workgroup
(WG), direct-to-LDS (DTLA/DTLB), prefetch depths (PGR / PLR), LDS layout.  The summary script below the run prints them per shape."""
import sys
import torch

shapes = [("L1 ff1", 49152, 5120, 640), ("L2 ff2", 12288, 1280, 5120), ("L1 conv3x3-K", 49152, 640, 5760), ("L2 qkv", 12288, 3840, 1280),
          ("4096^3", 4096, 4096, 4096)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16, generator=g) * 0.05
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        torch.mm(a, w.t(), out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        torch.mm(a, w.t(), out=o)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    print(f"SHAPE {name:14s} M={M} N={N} K={K}: torch.mm {us:.1f} us {2.0 * M * N * K / us / 1e6:.0f} TF", flush=True)
    del a, w, o
