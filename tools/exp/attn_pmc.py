"""One attention launch shape of level 0 a few times, for rocprofv3 --pmc passes: where do the cycles of attn_views_kernel go
(VALU busy / MFMA busy / waiting)?   python tools/exp/attn_pmc.py [intra|cross|text]   (the shipped variant choice: 4 waves x 2
query blocks, two workgroups per CU; logits with the spread the network's attention sees, as tools/exp/attn_ab.py)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "intra"
F, C, H, W = 16, 320, 32, 384
N, heads = H * W, C // 64
Nv = N // 6
INTRA = [[v] for v in range(6)]
CROSS = [[5, 1], [0, 2], [1, 3], [2, 4], [3, 5], [4]]
qk = (torch.randn(F * N, 2 * C, device="cuda") * 1.7).half()
vt = (torch.randn(F, C, N, device="cuda") * 0.5).half()
o = torch.empty(F * N, C, device="cuda", dtype=torch.float16)
kt, vtt = (torch.randn(2 * 80, C, device="cuda") * 1.7).half(), (torch.randn(2, C, 80, device="cuda") * 0.5).half()
for _ in range(6):
    if mode == "text":
        hip.attn_views(qk, 2 * C, kt, C, vtt, 80, C * 80, o, C, groups=F, heads=heads, H=H, W=W, views=1, kvH=1, kvW=80, kv_views=1,
                       kv_rows_per_group=80, q_per_kv=8, kv_valid=77, segs=[[0]], scale=0.125)
    else:
        hip.attn_views(qk, 2 * C, qk[:, C:], 2 * C, vt, N, C * N, o, C, groups=F, heads=heads, H=H, W=W, views=6, kvH=H, kvW=W,
                       kv_views=6, kv_rows_per_group=N, q_per_kv=1, kv_valid=Nv, segs=INTRA if mode == "intra" else CROSS, scale=0.125)
torch.cuda.synchronize()
