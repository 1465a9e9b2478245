"""Sample the shader clock (rocm-smi) while one GEMM shape runs back to back for a few seconds."""
import subprocess, sys, threading, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip

M, N, K = 12288, 1280, 5120
a = (torch.randn(M, K, device="cuda") * 0.5).half()
w = (torch.randn(N, K, device="cuda") * 0.5).half()
o = torch.empty(M, N, device="cuda", dtype=torch.float16)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            samples.append([l.strip() for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l])
        except Exception as e:
            samples.append([repr(e)])
        time.sleep(0.3)
t = threading.Thread(target=sampler); t.start()
time.sleep(1.0)
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        hip.gemm(a, w, M=M, N=N, K=K, lda=K, out16=o, ldc16=N)
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
stop = True; t.join()
print(f"{n} gemms in {dt:.2f}s -> {2.0*M*N*K*n/dt/1e12:.1f} TFLOP/s sustained")
for s in samples: print(s)
