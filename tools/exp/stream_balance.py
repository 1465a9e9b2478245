"""Which stream is the critical path of a two-stream evaluation?  Times (HIP events) from the start of the evaluation to
(a) the end of the ControlNet on the side stream and (b) the moment the UNet (main stream) reaches the join."""
import json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, configs, synth
from panacea_amd.nn import controlmodel as cm

kw = configs.get("full")
man = json.loads((ROOT / "tests/golden/manifest_full.json").read_text())
net = build_network(kw)
net.diffusion_model.load_state_dict(synth.synth_state_dict(man), strict=True)
net = net.to("cuda")
B, T, h, w = configs.SHAPES["full"]
g = {k: v.to("cuda") for k, v in synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"]).items()}
c = {k: g[k] for k in ("concat", "crossattn", "cond_feat")}
ev = {}
orig_rc = cm.ControlNet3D._run_control
def rc(self, rt, x16, hint, emb):
    out = orig_rc(self, rt, x16, hint, emb)
    ev["cn_end"] = torch.cuda.Event(enable_timing=True); ev["cn_end"].record(torch.cuda.current_stream())
    return out
cm.ControlNet3D._run_control = rc
orig_hs = cm.ControlNet3D._hint_stem
def hs(self, rt, hint):
    a = torch.cuda.Event(enable_timing=True); a.record(torch.cuda.current_stream())
    out = orig_hs(self, rt, hint)
    b = torch.cuda.Event(enable_timing=True); b.record(torch.cuda.current_stream())
    ev["hs"] = (a, b)
    return out
cm.ControlNet3D._hint_stem = hs
orig_ru = net.diffusion_model._run_unet
def ru(rt, x16, emb, control):
    def join():
        ev["join"] = torch.cuda.Event(enable_timing=True); ev["join"].record(torch.cuda.current_stream())
        return control()
    return orig_ru(rt, x16, emb, join if callable(control) else control)
net.diffusion_model._run_unet = ru
with torch.no_grad():
    for it in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); net(g["x"], g["t"], c); e.record()
        torch.cuda.synchronize()
        print(f"eval {it}: total {s.elapsed_time(e):.1f} ms | ControlNet done at {s.elapsed_time(ev['cn_end']):.1f} ms "
              f"(hint stem {ev['hs'][0].elapsed_time(ev['hs'][1]):.1f} ms) | UNet reaches the join at {s.elapsed_time(ev['join']):.1f} ms")
