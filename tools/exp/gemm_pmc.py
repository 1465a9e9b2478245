"""A few launches of the long-K GEMM shapes of config 3 for rocprofv3 --pmc passes: where do the cycles of the GEMM main loops go
(LDS port busy / bank conflicts / MFMA busy / waiting)?   python tools/exp/gemm_pmc.py [ff2|ff1|conv]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from panacea_amd import hip  # noqa: E402
from kbench import LEVELS, F, h16  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ff2"
C, H, W = LEVELS[2]
M = F * H * W
if mode == "ff2":          # M = 12288, N = 1280, K = 5120: gemm_glds_kernel 256x320, 80 K tiles, 990 TFLOP/s
    a, w = h16(M, 4 * C), h16(C, 4 * C) * 0.05
    res = torch.randn(M, C, device="cuda")
    o = torch.empty(M, C, device="cuda", dtype=torch.float16)
    fn = lambda: hip.gemm(a, w, M=M, N=C, K=4 * C, lda=4 * C, res1=res, ldr1=C, out16=o, ldc16=C)      # noqa: E731
elif mode == "ff1":        # M = 12288, N = 10240, K = 1280: persistent GEGLU kernel, staggered, 1110 TFLOP/s
    a, w = h16(M, C), h16(8 * C, C) * 0.1
    bias = torch.zeros(8 * C, device="cuda")
    o = torch.empty(M, 4 * C, device="cuda", dtype=torch.float16)
    fn = lambda: hip.gemm(a, w, M=M, N=8 * C, K=C, lda=C, bias=bias, geglu=True, out16=o, ldc16=4 * C)   # noqa: E731
else:                      # level-0 3x3 conv 320 -> 320 on the stencil-tile kernel, 1000 TFLOP/s
    C, H, W = LEVELS[0]
    M = F * H * W
    x, w = h16(F, H, W, C), h16(C, 9 * C) * 0.05
    o = torch.empty(M, C, device="cuda")
    fn = lambda: hip.gemm(x, w, M=M, N=C, K=9 * C, out32=o, ldc32=C, a_mode=hip.A_CONV3X3,                     # noqa: E731
                          conv=dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0))
for _ in range(6):
    fn()
torch.cuda.synchronize()
