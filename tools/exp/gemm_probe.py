"""Launch one GEMM-family shape a few times (for rocprofv3 --pmc passes).  usage: gemm_probe.py conv|plain M N K [iters]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
import os
ABL = int(os.environ.get("PNC_ABLATE", "0"))
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
h16 = lambda *s: (torch.randn(*s, device="cuda") * 0.5).half()
if kind == "conv":
    C = K // 9; F = 16; W = 96 * (M // (16 * 768)) if M >= 12288 else 48; H = M // (F * W)
    x, w = h16(F, H, W, C), h16(N, K)
    o = torch.empty(M, N, device="cuda")
    conv = dict(Cin=C, Hin=H, Win=W, Hout=H, Wout=W, stride=1, upsample=0)
    fn = lambda: hip.gemm(x, w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, out32=o, ldc32=N, act=ABL)
else:
    a, w = h16(M, K), h16(N, K)
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    fn = lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N, act=ABL)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
