"""debug: where do the LayerNorm variants of the persistent plain-A GEMM differ from the one-tile-per-workgroup kernel"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip  # noqa: E402
DEV = "cuda"
torch.manual_seed(0)
M, C = 256 * 512, 320
a = torch.randn(M, C, device=DEV).half()
w = (torch.randn(C, C, device=DEV) * C ** -0.5).half()
bias, gam, bet = torch.randn(C, device=DEV), torch.randn(C, device=DEV) * 0.3 + 1, torch.randn(C, device=DEV) * 0.3
res0 = torch.randn(M, C, device=DEV)


def run(opt, res):
    hip.set_option(hip.OPT_GEMM_PERSIST, opt)
    x = res0.clone() if res else torch.zeros(M, C, device=DEV)
    ln = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    kw = dict(res1=x, ldr1=C) if res else {}
    hip.gemm(a, w, M=M, N=C, K=C, lda=C, bias=bias, out32=x, ldc32=C, ln_gamma=gam, ln_beta=bet, ln_out16=ln, ldln=C, **kw)
    torch.cuda.synchronize()
    return x, ln


for res in (False, True):
    x1, l1 = run(1, res)
    x3, l3 = run(3, res)
    print("res", res, "x equal", torch.equal(x1, x3), "ln equal", torch.equal(l1, l3))
    d = (l1.float() - l3.float()).abs()
    bad = d > 0
    print("  ln mismatches", int(bad.sum()), "of", bad.numel(), "max", d.max().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  bad rows", rows.numel(), "first", rows[:16].tolist(), "rows mod 256:", sorted(set((rows % 256).tolist()))[:40])
        print("  bad cols", cols.numel(), "first", cols[:16].tolist())
        r0 = int(rows[0])
        print("  row", r0, "ref", l1[r0, :8].tolist(), "got", l3[r0, :8].tolist())
        ref_ln = torch.nn.functional.layer_norm(x1[r0], (C,), gam, bet, 1e-5)
        print("  torch LN of the row", ref_ln[:8].tolist())
hip.set_option(hip.OPT_GEMM_PERSIST, 3)
