import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
torch.manual_seed(0)
M, C = 512, 320
N, K = 8 * C, C
a = (torch.randn(M, K) * 1.0).half().cuda(); w = (torch.randn(N, K) * K ** -0.5).half().cuda(); bias = torch.randn(N).cuda()
o1 = torch.zeros(M, N // 2, device="cuda", dtype=torch.float16); o2 = torch.zeros_like(o1); lo = torch.zeros_like(o1)
hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o1, ldc16=N // 2)
hip.gemm(a, w, M=M, N=N, K=K, lda=K, bias=bias, geglu=True, out16=o2, ldc16=N // 2, out16_lo=lo)
torch.cuda.synchronize()
d = (o1.float() - o2.float())
bad = d != 0
print("mismatches", int(bad.sum()), "of", d.numel(), "max", d.abs().max().item())
idx = bad.nonzero()[:20]
for r, c in idx.tolist():
    print(r, c, o1[r, c].item(), o2[r, c].item())
print("rows mod 32 hist", torch.bincount(bad.nonzero()[:, 0] % 32, minlength=32).tolist())
print("cols mod 32 hist", torch.bincount(bad.nonzero()[:, 1] % 32, minlength=32).tolist())
# reference in fp64 of the ideal value to see which one is the correctly rounded
v = (a.double() @ w.double().T + bias.double())
pairs = v.view(M, N // 64, 2, 32)
ref = (pairs[:, :, 0] * torch.nn.functional.gelu(pairs[:, :, 1])).reshape(M, N // 2)
print("err new", (o1.double() - ref).abs().max().item(), "err old", (o2.double() - ref).abs().max().item())
