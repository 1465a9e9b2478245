"""Per-shape GEMM time of one full-size network evaluation (HIP events around every launch)."""
import json, sys, collections
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, configs, hip, synth

kw = configs.get("full")
man = json.loads((ROOT / "tests/golden/manifest_full.json").read_text())
net = build_network(kw)
net.diffusion_model.load_state_dict(synth.synth_state_dict(man), strict=True)
net = net.to("cuda")
net.diffusion_model.two_stream = False
if len(sys.argv) > 1:
    net.diffusion_model.precision = sys.argv[1]          # fast | precise | precise-f16lo | precise-all
B, T, h, w = configs.SHAPES["full"]
g = {k: v.to("cuda") for k, v in synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"]).items()}
c = {k: g[k] for k in ("concat", "crossattn", "cond_feat")}
rec = []
orig = hip.gemm
def gemm(a16, w16, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(a16, w16, **k); e1.record()
    tag = ("geglu " if k.get("geglu") else "") + ("res " if k.get("res1") is not None else "") + ("T " if k.get("out16t") is not None else "") + ("o32 " if k.get("out32") is not None else "") + ("o16" if k.get("out16") is not None else "") + ("" if k.get("a16_lo") is None else (" lo8" if k["a16_lo"].dtype == torch.uint8 else " lo16")) + (" ln" if k.get("ln_out16") is not None else "")
    rec.append(((k.get("a_mode", 0), k["M"], k["N"], k["K"], tag), e0, e1))
with torch.no_grad():
    net(g["x"], g["t"], c)
    hip.gemm = gemm
    import panacea_amd.engine as E
    net(g["x"], g["t"], c)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total gemm-family {tot:.1f} ms over {len(rec)} launches")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    mode, M, N, K, tag = key
    fl = 2.0 * M * N * K * n
    print(f"mode{mode} M={M:7d} N={N:5d} K={K:5d} {tag:26s} x{n:3d}  {ms:7.2f} ms  {ms/n*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF")
