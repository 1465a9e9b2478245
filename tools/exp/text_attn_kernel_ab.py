import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import torch
from panacea_amd import hip
import emu
from kbench import LEVELS, F, timeit
for li,(C,H,W) in enumerate(LEVELS):
    N, heads, M = H*W, C//64, F*H*W
    g = torch.Generator(device="cuda").manual_seed(li)
    q = (torch.randn(M, C, device="cuda", generator=g)*1.7).half()
    kt = (torch.randn(2*80, C, device="cuda", generator=g)*1.7).half(); kt.view(2,80,C)[:,77:] = 0
    vtt = (torch.randn(2, C, 80, device="cuda", generator=g)*0.5).half(); vtt[:,:,77:] = 0
    o = torch.empty(M, C, device="cuda", dtype=torch.float16)
    kw = dict(groups=F, heads=heads, H=H, W=W, views=1, kvH=1, kvW=80, kv_views=1, kv_rows_per_group=80, q_per_kv=8, kv_valid=77, segs=[[0]], scale=0.125)
    fn = lambda: hip.attn_views(q, C, kt, C, vtt, 80, C*80, o, C, **kw)
    oe = torch.empty_like(o)
    if li >= 2:
        emu.attn_views(q, C, kt, C, vtt, 80, C*80, oe, C, **kw)
    res = {}
    for rd in range(2):
        for v in (42, 0):
            hip.set_option(hip.OPT_ATTN_VARIANT, v)
            t = timeit(fn, iters=16, warm=3)
            res[v] = o.clone()
            err = (o.float()-oe.float()).abs().max().item() if li >= 2 else float('nan')
            print(f"text L{li} round {rd} {'attn_views 4x2' if v else 'attn_text     '}: {t*1e6:7.1f} us  {M*C*4/t/1e9:6.0f} GB/s  err vs emu {err:.2e}", flush=True)
    print("   max |text - views|", (res[0].float()-res[42].float()).abs().max().item())
hip.set_option(hip.OPT_ATTN_VARIANT, 0)
