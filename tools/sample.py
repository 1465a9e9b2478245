#!/usr/bin/env python
"""End-to-end demo of everything this package owns, with SYNTHETIC weights (no checkpoint, no dataset offline):
conditioning tensors -> N Euler/CFG steps of the ControlNet-UNet (step invariants hoisted) -> first-stage decode ->
per-view JPEGs + panorama GIF.  With `--ckpt <engine checkpoint>` the reference's weights are loaded instead
(`model.diffusion_model.*` into the denoiser, `first_stage_model.*` into the decoder).

    python tools/sample.py --steps 5 --out gpurun_out/sample
"""
import argparse, json, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, checkpoint, configs, pipeline, synth   # noqa: E402
from panacea_amd.nn import model                                               # noqa: E402

VAE = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
           num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--out", default="gpurun_out/sample")
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--seed", type=int, default=3407)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    kw = configs.get("full")
    net = build_network(kw)
    fs = model.FirstStageDecoder(4, VAE)
    if a.ckpt:
        print(checkpoint.load_denoiser(net, a.ckpt))
        sd = checkpoint.denoiser_state_dict(checkpoint.read_state_dict(a.ckpt), "first_stage_model.")
        print(fs.load_state_dict(sd, strict=False))
    else:
        man = json.loads((ROOT / "tests/golden/manifest_full.json").read_text())
        net.diffusion_model.load_state_dict(synth.synth_state_dict(man), strict=True)
        fs.load_state_dict(synth.synth_state_dict({k: list(v.shape) for k, v in fs.state_dict().items()}), strict=True)
    net, fs = net.to(dev), fs.to(dev)
    B, T, h, w = configs.SHAPES["full"]
    g = {k: v.to(dev) for k, v in synth.synth_inputs(2, T, h, w, context_dim=kw["context_dim"]).items()}
    cond = {"crossattn": g["crossattn"][1:2], "concat": g["concat"][T:], "cond_feat": g["cond_feat"][T:]}
    uc = {"crossattn": g["crossattn"][0:1], "concat": g["concat"][:T], "cond_feat": g["cond_feat"][:T]}
    noise = torch.randn(T, 4, h, w, generator=torch.Generator().manual_seed(a.seed)).to(dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames = pipeline.sample_frames(net, fs, cond, uc, noise, num_steps=a.steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{a.steps} steps + decode of {T} frames {tuple(frames.shape)}: {dt:.2f} s "
          f"(range {frames.min().item():.2f} .. {frames.max().item():.2f}, finite={bool(torch.isfinite(frames).all())})")
    out = Path(a.out); out.mkdir(parents=True, exist_ok=True)
    checkpoint.save_view_frames(frames, str(out), [f"scene__{v}__000" for v in range(6)], view_width=frames.shape[-1] // 6)
    checkpoint.save_gif(frames[:, :, ::4, ::4], str(out / "panorama_quarter.gif"))
    print("wrote", out)


if __name__ == "__main__":
    main()
