"""Acceptance fixture of SURVEY.md section 8 f1 / VERDICT r5 item 6c (build container only): the REFERENCE's own sampler stack

    EulerEDMSampler.__call__ / sampler_step     sgm/modules/diffusionmodules/sampling.py:112-133,214-218
    DiscreteDenoiser.__call__                   .../denoiser.py:22-28
    VanillaCFG                                  .../guiders.py:25-40

drives, for 3 steps on the tiny Panacea configuration (2 levels, intra-view + cross-view + temporal attention, ControlNet),
  (i)  the reference network (OpenAIWrapperControlLDM3D around ControlledUNetModel3D + ControlNet3D), and
  (ii) the classes `panacea_amd.dropin.install()` puts in their place — built through the reference's own
       `instantiate_from_config`, run here on the CPU emulation of the C-ABI (tests/emu.py; the GPU is not in this container).
(i) is committed as tests/golden/sampler_tiny_net.npz: the latent after every sampler step and the timestep indices the network
saw.  (ii) must stay within the stated fp16 tolerance of (i) at every step before the file is written; the same trajectory is
replayed on the MI355X through panacea_amd.sampling by tests/test_sampler_acceptance.py (-m gpu).

    python -m oracle.gen_golden_sampler_net
"""
from __future__ import annotations

import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import ref_import                                  # noqa: E402
from oracle.gen_golden import GOLDEN                           # noqa: E402
from panacea_amd import configs, synth                         # noqa: E402

P = "sgm.modules.diffusionmodules."
STEPS, CFG_SCALE = 3, 5.0
TOL = 1e-3          # |x_dropin - x_reference| per step, relative to max|x| (measured 3.4e-4: the emulated fp16 operand path over 3 network evaluations)


def reference_stack(ns):
    for m in ("guiders", "discretizer", "denoiser_scaling", "denoiser_weighting", "sampling_utils"):
        importlib.import_module(P + m)
    disc = {"target": P + "discretizer.LegacyDDPMDiscretization"}
    den = ns.dn.DiscreteDenoiser(weighting_config={"target": P + "denoiser_weighting.EpsWeighting"},
                                 scaling_config={"target": P + "denoiser_scaling.EpsScaling"}, num_idx=1000,
                                 discretization_config=disc)
    smp = ns.sp.EulerEDMSampler(num_steps=STEPS, discretization_config=disc, device="cpu",
                                guider_config={"target": P + "guiders.VanillaCFG", "params": {"scale": CFG_SCALE}})
    return den, smp


def sampler_inputs(kw):
    """one sample (T frames) with its conditional / unconditional conditioning: the second / first sample of the tiny step inputs"""
    B, T, h, w = configs.SHAPES["tiny"]
    inp = synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"])
    c = {"crossattn": inp["crossattn"][1:2].clone(), "concat": inp["concat"][T:].clone(), "cond_feat": inp["cond_feat"][T:].clone()}
    uc = {"crossattn": inp["crossattn"][0:1].clone(), "concat": inp["concat"][:T].clone(), "cond_feat": inp["cond_feat"][:T].clone()}
    return inp["x"][T:].clone(), c, uc


def run(smp, den, network, x0, c, uc):
    """the reference sampler's own loop; records x after every sampler_step and the timestep indices handed to the network"""
    xs, seen = [], []
    step = smp.sampler_step

    def recording_step(*a, **k):
        x = step(*a, **k)
        xs.append(x.detach().clone())
        return x
    smp.sampler_step = recording_step

    def net(x, t, cc):
        seen.append(t.clone())
        return network(x, t, cc)
    try:
        with torch.no_grad():
            out = smp(lambda inp, sigma, cc: den(net, inp, sigma, cc), x0.clone(), c, uc)
    finally:
        smp.sampler_step = step
    assert torch.equal(out, xs[-1])
    return torch.stack(xs), torch.stack(seen)[:, 0]


def main():
    ns = ref_import.import_reference()
    kw = configs.get("tiny")
    den, smp = reference_stack(ns)
    # (i) the reference network
    net, wrapper = ref_import.build_reference_network(ns, kw)
    manifest = {k: list(v.shape) for k, v in net.state_dict().items()}
    sd = synth.synth_state_dict(manifest)
    net.load_state_dict(sd, strict=True)
    x0, c, uc = sampler_inputs(kw)
    xs_ref, t_ref = run(smp, den, wrapper, x0, dict(c), dict(uc))
    print(f"reference: timesteps {t_ref.tolist()}  |x| after each step {[round(float(x.abs().max()), 3) for x in xs_ref]}")
    # (ii) the dropped-in classes, built by the reference's own instantiate_from_config from the YAML target strings
    import emu
    from panacea_amd import dropin, engine as E
    dropin.install()
    cm = sys.modules["sgm.modules.diffusionmodules.controlmodel"]
    wr = sys.modules["sgm.modules.diffusionmodules.wrappers"]
    cn_cfg = {"target": "sgm.modules.diffusionmodules.controlmodel.ControlNet3D",
              "params": dict(kw, hint_channels=19, control_scales=1.0)}
    mirror = ns.util.instantiate_from_config({"target": "sgm.modules.diffusionmodules.controlmodel.ControlledUNetModel3D",
                                              "params": dict(kw, controlnet_config=cn_cfg, out_channels=4)}).eval()
    assert type(mirror).__module__.startswith("panacea_amd"), "the drop-in did not take"
    mirror.load_state_dict(sd, strict=True)
    mwrap = wr.OpenAIWrapperControlLDM3D(mirror)
    assert type(mwrap).__module__.startswith("panacea_amd") and cm._panacea_amd
    with E.use_backend(emu):
        xs_mir, t_mir = run(smp, den, mwrap, x0, dict(c), dict(uc))
    assert torch.equal(t_ref, t_mir)
    errs = [(a - b).abs().max().item() / a.abs().max().item() for a, b in zip(xs_ref, xs_mir)]
    print("drop-in (emulated kernels) vs reference, max-abs / max|x| per step:", [f"{e:.2e}" for e in errs])
    assert max(errs) <= TOL, errs
    np.savez_compressed(GOLDEN / "sampler_tiny_net.npz", x_steps=xs_ref.numpy(), timesteps=t_ref.numpy(), x0=x0.numpy(),
                        steps=np.int32(STEPS), cfg_scale=np.float32(CFG_SCALE), tol_rel=np.float32(TOL),
                        dropin_emu_err_rel=np.asarray(errs, dtype=np.float32))
    print("written tests/golden/sampler_tiny_net.npz")


if __name__ == "__main__":
    main()
