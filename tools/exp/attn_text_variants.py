"""text cross-attention (77 keys, a stream-bound launch: q in, o out) on the four workgroup shapes of attn_views_kernel"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip
from tools.kbench import timeit, h16, LEVELS, F

for li, (C, H, W) in enumerate(LEVELS):
    M, heads = F * H * W, C // 64
    q, o = h16(M, C), torch.empty(M, C, device="cuda", dtype=torch.float16)
    kt, vtt = h16(2 * 80, C), h16(2, C, 80)
    fn = lambda: hip.attn_views(q, C, kt, C, vtt, 80, C * 80, o, C, groups=F, heads=heads, H=H, W=W, views=1, kvH=1, kvW=80,
                                kv_views=1, kv_rows_per_group=80, q_per_kv=8, kv_valid=77, segs=[[0]], scale=0.125)
    res = []
    for v in (0, 82, 81, 42, 41, 0):
        prev = hip.set_option(hip.OPT_ATTN_VARIANT, v)
        try:
            res.append((v, timeit(fn, iters=20, warm=3)))
        finally:
            hip.set_option(hip.OPT_ATTN_VARIANT, prev)
    print(f"text attention L{li} M={M} heads={heads}: " + "  ".join(f"v{v}={t * 1e6:6.1f}us" for v, t in res), flush=True)
