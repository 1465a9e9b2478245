"""Which host-side ops of one fused denoising step end up as device memcpys (`__amd_rocclr_copyBuffer` in the rocprofv3 stats:
~200 per step)?  torch.profiler with python stacks, memcpy / memset events grouped by the innermost panacea_amd frame."""
import collections
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import product_network, step_inputs  # noqa: E402
from panacea_amd import configs, sampling  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
kw = configs.get(name)
T = kw["num_frames"]
net, _, _ = product_network(name, device="cuda")
g = step_inputs(name, kw, device="cuda", t_index=500)
cond = {"crossattn": g["crossattn"][1:2], "concat": g["concat"][T:], "cond_feat": g["cond_feat"][T:]}
uc = {"crossattn": g["crossattn"][0:1], "concat": g["concat"][:T], "cond_feat": cond["cond_feat"]}
den = sampling.DiscreteDenoiser().to("cuda")
smp = sampling.EulerEDMSampler(50, guider=sampling.VanillaCFG(5.0), device="cuda")
sig = smp.sigmas()
x = g["x"][T:] * torch.sqrt(1.0 + sig[0] ** 2.0)
s_in = x.new_ones([x.shape[0]])
denoiser = sampling.BoundDenoiser(den, net)
with torch.no_grad():
    for _ in range(2):
        smp.sampler_step(s_in * sig[0], s_in * sig[1], denoiser, x, cond, uc)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        smp.sampler_step(s_in * sig[0], s_in * sig[1], denoiser, x, cond, uc)
        torch.cuda.synchronize()
by = collections.Counter()
names = collections.Counter()
for ev in prof.events():
    n = ev.name
    if "emcpy" in n or "emset" in n or "copy_" == n or n in ("aten::copy_", "aten::zero_", "aten::fill_"):
        names[n] += 1
        frames = [f for f in (ev.stack or []) if "panacea_amd" in f or "bench" in f]
        by[(n, frames[0] if frames else (ev.stack[0] if ev.stack else "?"))] += 1
print(names.most_common(12))
for (n, fr), c in by.most_common(40):
    print(f"{c:5d}  {n:28s} {fr}")
