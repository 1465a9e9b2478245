import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import torch
from panacea_amd import hip
from kbench import LEVELS, F, timeit, h16
for li,(C,H,W) in enumerate(LEVELS[:3]):
    N, heads, M = H*W, C//64, F*H*W
    qk, o = h16(M, 2*C), torch.empty(M, C, device="cuda", dtype=torch.float16)
    kt, vtt = h16(2*80, C), h16(2, C, 80)
    fn = lambda: hip.attn_views(qk, 2*C, kt, C, vtt, 80, C*80, o, C, groups=F, heads=heads, H=H, W=W, views=1, kvH=1, kvW=80, kv_views=1, kv_rows_per_group=80, q_per_kv=8, kv_valid=77, segs=[[0]], scale=0.125)
    for rd in range(2):
        for v in (41, 42, 81, 82):
            hip.set_option(hip.OPT_ATTN_VARIANT, v)
            t = timeit(fn, iters=16, warm=3)
            print(f"text L{li} round {rd} variant {v}: {t*1e6:7.1f} us  {M*C*4/t/1e9:6.0f} GB/s", flush=True)
hip.set_option(hip.OPT_ATTN_VARIANT, 0)
