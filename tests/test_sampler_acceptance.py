"""Acceptance test of the drop-in boundary: a TRAJECTORY, not a class rebinding.

tests/golden/sampler_tiny_net.npz holds the latent after each of 3 sampler steps when the REFERENCE's own sampler stack —
EulerEDMSampler (sampling.py:112-133,214-218), DiscreteDenoiser (denoiser.py:22-28), VanillaCFG(5) (guiders.py:25-40) — drives the
REFERENCE network on the tiny Panacea configuration (oracle/gen_golden_sampler_net.py, generated from the imported reference).

  * build container (needs /root/reference): the reference's sampler stack drives the classes `panacea_amd.dropin.install()` rebinds,
    built by the reference's own `instantiate_from_config`, on the emulated kernels -> the golden trajectory;
  * anywhere (CPU): panacea_amd.sampling's mirrors drive the product network on the emulated kernels -> the golden trajectory;
  * -m gpu: the same replay on the MI355X through the C-ABI, plain step, fused step and hoisted invariants.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from helpers import GOLDEN, product_network
from panacea_amd import configs, sampling as S, synth

G = np.load(GOLDEN / "sampler_tiny_net.npz")
TOL = float(G["tol_rel"])                     # per step, relative to max|x| of the reference latent


def _inputs(kw, device="cpu"):
    B, T, h, w = configs.SHAPES["tiny"]
    inp = synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"])
    c = {"crossattn": inp["crossattn"][1:2].clone(), "concat": inp["concat"][T:].clone(), "cond_feat": inp["cond_feat"][T:].clone()}
    uc = {"crossattn": inp["crossattn"][0:1].clone(), "concat": inp["concat"][:T].clone(), "cond_feat": inp["cond_feat"][:T].clone()}
    x0 = inp["x"][T:].clone()
    assert np.array_equal(x0.numpy(), G["x0"])
    mv = lambda d: {k: v.to(device) for k, v in d.items()}      # noqa: E731
    return x0.to(device), mv(c), mv(uc)


def _trajectory(smp, denoiser, x0, c, uc, **kw):
    """EulerEDMSampler.__call__ of panacea_amd.sampling with the latent recorded after every step"""
    xs = []
    step = smp.sampler_step

    def rec(*a, **k):
        x = step(*a, **k)
        xs.append(x.detach().float().cpu().clone())
        return x
    smp.sampler_step = rec
    try:
        with torch.no_grad():
            smp(denoiser, x0.clone(), c, uc, **kw)
    finally:
        smp.sampler_step = step
    return torch.stack(xs)


def _check(xs, tag):
    ref = torch.from_numpy(G["x_steps"])
    assert xs.shape == ref.shape
    errs = [((a - b).abs().max() / b.abs().max()).item() for a, b in zip(xs, ref)]
    from helpers import measured
    measured(f"sampler_acceptance_{tag}", **{f"step{i}": e for i, e in enumerate(errs)})
    assert max(errs) <= TOL, (tag, errs)
    return errs


def test_sampling_mirrors_replay_the_reference_trajectory_on_the_emulation():
    import emu
    from panacea_amd import engine as E
    w, _, kw = product_network("tiny")
    x0, c, uc = _inputs(kw)
    den = S.DiscreteDenoiser()
    smp = S.EulerEDMSampler(int(G["steps"]), guider=S.VanillaCFG(float(G["cfg_scale"])), device="cpu")
    assert S.timestep_indices(int(G["steps"])) == G["timesteps"].tolist()
    with E.use_backend(emu):
        xs = _trajectory(smp, S.BoundDenoiser(den, w), x0, c, uc)
    _check(xs, "emu")


def test_reference_sampler_stack_drives_the_dropin_network():
    """build container only: the reference's OWN EulerEDMSampler / DiscreteDenoiser / VanillaCFG objects call the dropped-in
    OpenAIWrapperControlLDM3D — the way inference.py would — and land on the trajectory they produce around the reference network"""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("/root/reference is not present (build container only)")
    import emu
    from oracle import gen_golden_sampler_net as gen
    from panacea_amd import dropin, engine as E
    ns = ref_import.import_reference()
    den, smp = gen.reference_stack(ns)
    kw = configs.get("tiny")
    dropin.install()
    wr = sys.modules["sgm.modules.diffusionmodules.wrappers"]
    cn_cfg = {"target": "sgm.modules.diffusionmodules.controlmodel.ControlNet3D", "params": dict(kw, hint_channels=19, control_scales=1.0)}
    mirror = ns.util.instantiate_from_config({"target": "sgm.modules.diffusionmodules.controlmodel.ControlledUNetModel3D",
                                              "params": dict(kw, controlnet_config=cn_cfg, out_channels=4)}).eval()
    assert type(mirror).__module__.startswith("panacea_amd")
    from helpers import manifest
    mirror.load_state_dict(synth.synth_state_dict(manifest("tiny")), strict=True)
    x0, c, uc = gen.sampler_inputs(kw)
    with E.use_backend(emu):
        xs, ts = gen.run(smp, den, wr.OpenAIWrapperControlLDM3D(mirror), x0, c, uc)
    assert ts.tolist() == G["timesteps"].tolist()
    _check(xs, "reference_stack_emu")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "fused", "fused+hoisted"])
def test_reference_trajectory_replayed_on_the_gpu(mode):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    w, _, kw = product_network("tiny", device="cuda")
    x0, c, uc = _inputs(kw, "cuda")
    den = S.DiscreteDenoiser().to("cuda")
    smp = S.EulerEDMSampler(int(G["steps"]), guider=S.VanillaCFG(float(G["cfg_scale"])), device="cuda")
    smp.fuse = mode != "plain"
    bd = S.BoundDenoiser(den, w)
    assert smp._fusable(bd, x0, c) == (mode != "plain")
    xs = _trajectory(smp, bd, x0, c, uc, **({"network": w} if mode == "fused+hoisted" else {}))
    _check(xs, "gpu_" + mode.replace("+", "_"))
    assert "libpanacea_hip.so" in Path("/proc/self/maps").read_text()
