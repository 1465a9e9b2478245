"""One attention launch shape (level-0 intra-view, the 8 x 2 variant) a few times, for rocprofv3 --pmc passes:
where do the cycles of attn_views_kernel go (VALU busy / MFMA busy / waiting)?   python tools/exp/attn_pmc.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip  # noqa: E402

F, C, H, W = 16, 320, 32, 384
N, heads = H * W, C // 64
Nv = N // 6
INTRA = [[v] for v in range(6)]
qk = (torch.randn(F * N, 2 * C, device="cuda") * 0.5).half()
vt = (torch.randn(F, C, N, device="cuda") * 0.5).half()
o = torch.empty(F * N, C, device="cuda", dtype=torch.float16)
for _ in range(6):
    hip.attn_views(qk, 2 * C, qk[:, C:], 2 * C, vt, N, C * N, o, C, groups=F, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W,
                   kv_views=6, kv_rows_per_group=N, q_per_kv=1, kv_valid=Nv, segs=INTRA, scale=0.125)
torch.cuda.synchronize()
