"""The library's plain fp16-output GEMM against the vendor GEMM behind torch.mm (hipBLASLt / rocBLAS) on the transformer shapes of config 3:
same operands, fp16 out, no epilogue work on either side."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
from tools.kbench import timeit, h16

for name, M, N, K in (("L0 qkv", 196608, 960, 320), ("L0 proj", 196608, 320, 320), ("L0 ff1 (no gate)", 196608, 2560, 320),
                      ("L0 ff2", 196608, 320, 1280), ("L1 qkv", 49152, 1920, 640), ("L1 ff1 (no gate)", 49152, 5120, 640),
                      ("L2 qkv", 12288, 3840, 1280), ("L2 ff2", 12288, 1280, 5120)):
    a, w = h16(M, K), h16(N, K)
    o = torch.empty(M, N, device="cuda", dtype=torch.float16)
    wt = w.t()
    t_lib = timeit(lambda: hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N), iters=16, warm=3)
    t_ven = timeit(lambda: torch.mm(a, wt, out=o), iters=16, warm=3)
    fl = 2.0 * M * N * K
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:5d}: library {t_lib * 1e6:7.1f} us ({fl / t_lib / 1e12:6.1f} TF)   torch.mm {t_ven * 1e6:7.1f} us ({fl / t_ven / 1e12:6.1f} TF)",
          flush=True)
    del a, w, o
