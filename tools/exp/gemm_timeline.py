"""s_memtime timeline of one mid-grid workgroup of a GEMM (act bit 0x20000).  usage: gemm_timeline.py M N K [geglu|res]

Needs the instrumented kernel of commit 781529f (the stamps cost ~5 ms/step in the product kernels through register
pressure in the epilogue and were removed again); its output is kept in profiles/round1/gemm_timeline_r1i.txt."""
import ctypes, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
lib = hip.load()
a = (torch.randn(M, K, device="cuda") * 0.5).half(); w = (torch.randn(N, K, device="cuda") * 0.5).half()
bias = torch.zeros(N, device="cuda")
ws = torch.zeros(64, device="cuda")
p = hip.GemmParams()
p.A, p.W, p.M, p.N, p.K, p.lda = a.data_ptr(), w.data_ptr(), M, N, K, K
p.bias = bias.data_ptr(); p.n_split = N
if mode == "geglu":
    o = torch.empty(M, N // 2, device="cuda", dtype=torch.float16); p.out16, p.ldc16, p.geglu = o.data_ptr(), N // 2, 1
elif mode == "res":
    o = torch.zeros(M, N, device="cuda"); p.res1, p.ldr1, p.out32, p.ldc32 = o.data_ptr(), N, o.data_ptr(), N
else:
    o = torch.empty(M, N, device="cuda", dtype=torch.float16); p.out16, p.ldc16 = o.data_ptr(), N
p.ws, p.ws_floats = ws.data_ptr(), 0          # ws_floats = 0: never split K; the pointer is only the dump target
st = torch.cuda.current_stream().cuda_stream
for it in range(3):
    p.act = 0x20000
    assert lib.pnc_gemm_f16(ctypes.byref(p), ctypes.c_void_p(st)) == 0
torch.cuda.synchronize()
r = ws.view(8, 8).cpu()
print(f"M={M} N={N} K={K} {mode}   (shader cycles of one workgroup; MFMA-only time of the K loop = ktiles x 1024 per wave)")
print("wave   total  prologue     issue   compute     vmcnt   barrier  epilogue  ktiles")
for wv in range(8):
    print(f"{wv:4d} " + " ".join(f"{x:9.0f}" for x in r[wv].tolist()))
