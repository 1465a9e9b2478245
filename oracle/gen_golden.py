"""Generate tests/golden/ from the REFERENCE ITSELF (build container only).

    python -m oracle.gen_golden

For each small configuration: build the reference network (oracle/ref_import.py), load the
deterministic synthetic weights (panacea_amd/synth.py — names+shapes only, zero-init tensors made
non-zero), run `OpenAIWrapperControlLDM3D.forward` on synthetic inputs under no_grad, and store
  * eps (full) and stride-7 samples of the ControlNet residuals and of every top-level block output
                                                                     -> tests/golden/<cfg>.npz (fp32)
  * the state-dict manifest name -> shape                           -> tests/golden/manifest_<cfg>.json
The full Panacea+ network only contributes its manifest (2 478 tensors; built on the meta device).
The oracle is checked against the same run before anything is written (max-abs <= 2e-5).
Only data is written: no reference source, bytecode or pickled module goes into the repo.
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import panacea_oracle as po            # noqa: E402
from oracle import ref_import                      # noqa: E402
from panacea_amd import configs, synth             # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def oracle_cfg(kw: dict) -> po.OracleConfig:
    return po.OracleConfig(num_frames=kw["num_frames"], model_channels=kw["model_channels"],
                           num_head_channels=kw["num_head_channels"],
                           spatial_only_attn_type=kw["spatial_only_attn_type"],
                           insert_crossview=kw["insert_crossview"])


def run_config(ns, name: str, shape=None, out_name=None, eps_only=False):
    """`shape` / `out_name`: the same network on another latent size (plain64 = BASELINE config 1 as written: single view,
    one frame, 64x64 latent); `eps_only` stores eps alone."""
    kw = configs.get(name)
    B, T, h, w = shape or configs.SHAPES[name]
    net, wrapper = ref_import.build_reference_network(ns, kw)
    manifest = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(manifest)
    net.load_state_dict(sd, strict=True)
    inp = synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"])

    trace = {}

    def hook(prefix):
        def fn(_m, _i, out):
            trace[prefix] = out.detach().clone()
        return fn
    for i, m in enumerate(net.input_blocks):
        m.register_forward_hook(hook(f"input_blocks.{i}"))
    net.middle_block.register_forward_hook(hook("middle_block"))
    for i, m in enumerate(net.output_blocks):
        m.register_forward_hook(hook(f"output_blocks.{i}"))
    for i, m in enumerate(net.controlnet.input_blocks):
        m.register_forward_hook(hook(f"controlnet.input_blocks.{i}"))
    net.controlnet.middle_block.register_forward_hook(hook("controlnet.middle_block"))
    net.controlnet.input_hint_block.register_forward_hook(hook("controlnet.input_hint_block"))
    controls = {}

    def cn_hook(_m, _i, out):
        for j, c in enumerate(out):
            controls[f"control.{j}"] = c.detach().clone()
    net.controlnet.register_forward_hook(cn_hook)

    c = {"concat": inp["concat"].clone(), "crossattn": inp["crossattn"].clone(), "cond_feat": inp["cond_feat"].clone()}
    with torch.no_grad():
        eps = wrapper(inp["x"].clone(), inp["t"].clone(), c)

    # ---- pin the oracle on the very same run ----
    po.TRACE = {}
    eps_o = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"],
                               {k: inp[k] for k in ("concat", "crossattn", "cond_feat")})
    otrace, po.TRACE = po.TRACE, None
    d = (eps - eps_o).abs().max().item()
    print(f"[{name}] eps rms {eps.pow(2).mean().sqrt():.4f} max {eps.abs().max():.4f}; oracle vs reference max-abs {d:.3e}")
    assert d <= 2e-5, "oracle disagrees with the reference"
    for k, v in trace.items():
        if k == "controlnet.input_hint_block":
            continue
        dk = (v - otrace[k]).abs().max().item()
        assert dk <= 5e-5 * max(1.0, v.abs().max().item()), (k, dk)

    out = {"eps": eps.numpy()}
    # everything but eps: a fixed stride-7 sample of the flattened tensor keeps the fixture small
    if not eps_only:
        out.update({k: v.reshape(-1)[::7].numpy() for k, v in controls.items()})
        out.update({"block." + k: v.reshape(-1)[::7].numpy() for k, v in trace.items()})
    GOLDEN.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLDEN / f"{out_name or name}.npz", **out)
    if out_name is None:
        (GOLDEN / f"manifest_{name}.json").write_text(json.dumps(manifest, indent=0, sort_keys=True))
    print(f"[{name}] wrote {len(out)} arrays, {len(manifest)} manifest entries")


def full_manifest(ns):
    net, _ = ref_import.build_reference_network(ns, configs.get("full"), device="meta")
    manifest = {k: list(v.shape) for k, v in net.state_dict().items()}
    nparam = sum(int(np.prod(v)) for v in manifest.values())
    (GOLDEN / "manifest_full.json").write_text(json.dumps(manifest, indent=0, sort_keys=True))
    print(f"[full] {len(manifest)} tensors, {nparam / 1e6:.1f} M parameters")


def small_vectors(ns):
    """Known-answer vectors of the two closed-form tables on the path."""
    t = torch.tensor([999, 959, 666, 333, 39, 0], dtype=torch.int64)
    out = {"timestep_embedding.t": t.numpy(),
           "timestep_embedding.320": ns.dutil.timestep_embedding(t, 320).numpy()}
    for T, C in [(8, 320), (2, 64), (8, 1280)]:
        out[f"pos_embed.{T}.{C}"] = ns.att.create_1d_absolute_sin_cos_embedding(T, C).numpy()
    np.savez_compressed(GOLDEN / "tables.npz", **out)
    print("[tables] written")


def fake_network(x, t, c):
    """closed-form stand-in for eps_theta so that the sampler / denoiser / guider arithmetic can be pinned"""
    return torch.tanh(0.3 * x) * 0.5 + 1e-4 * t.float()[:, None, None, None] + 0.01 * c["crossattn"].mean() \
        + 0.05 * c["concat"]


def sampler_vectors(ns):
    """3- and 25-step EulerEDMSampler + VanillaCFG(5) + DiscreteDenoiser(EpsScaling) trajectory of the REFERENCE
    classes (sampling.py, guiders.py, denoiser.py, discretizer.py) around `fake_network`."""
    import importlib
    P = "sgm.modules.diffusionmodules."
    for m in ("guiders", "discretizer", "denoiser_scaling", "denoiser_weighting", "sampling_utils"):
        importlib.import_module(P + m)
    disc = {"target": P + "discretizer.LegacyDDPMDiscretization"}
    den = ns.dn.DiscreteDenoiser(weighting_config={"target": P + "denoiser_weighting.EpsWeighting"},
                                 scaling_config={"target": P + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                                 discretization_config=disc)
    out = {"denoiser.sigmas": den.sigmas.numpy()}
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 4, 4, 12, generator=g)
    c = {"crossattn": torch.randn(1, 77, 8, generator=g), "concat": torch.randn(2, 4, 4, 12, generator=g),
         "cond_feat": torch.rand(2, 19, 8, 8, generator=g)}
    uc = {"crossattn": torch.randn(1, 77, 8, generator=g), "concat": c["concat"].clone(), "cond_feat": c["cond_feat"].clone()}
    for n in (3, 25):
        smp = ns.sp.EulerEDMSampler(num_steps=n, discretization_config=disc, device="cpu",
                                    guider_config={"target": P + "guiders.VanillaCFG", "params": {"scale": 5.0}})
        seen = []

        def net(x, t, cc):
            seen.append(t.clone())
            return fake_network(x, t, cc)
        xs = smp(lambda inp, sigma, cc: den(net, inp, sigma, cc), x0.clone(), c, uc)
        out[f"sampler.{n}.sigmas"] = smp.discretization(n, device="cpu").numpy()
        out[f"sampler.{n}.timesteps"] = torch.stack(seen)[:, 0].numpy()
        out[f"sampler.{n}.x_final"] = xs.numpy()
    np.savez_compressed(GOLDEN / "sampler.npz", **out)
    print("[sampler] timesteps(3) =", out["sampler.3.timesteps"], " sigma0 =", out["sampler.25.sigmas"][0])


if __name__ == "__main__":
    torch.manual_seed(0)
    ns = ref_import.import_reference()
    if "--sampler-only" in sys.argv:
        sampler_vectors(ns)
        sys.exit(0)
    if "--plain64-only" in sys.argv:
        run_config(ns, "plain1", shape=(1, 1, 64, 64), out_name="plain64", eps_only=True)
        sys.exit(0)
    small_vectors(ns)
    sampler_vectors(ns)
    for name in ("tiny", "plain1"):
        run_config(ns, name)
    run_config(ns, "plain1", shape=(1, 1, 64, 64), out_name="plain64", eps_only=True)     # BASELINE config 1 as written
    full_manifest(ns)
