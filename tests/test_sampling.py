"""Sampler / denoiser / guider mirrors against a trajectory of the reference's own
classes around a closed-form stand-in network (tests/golden/sampler.npz, oracle/gen_golden.py)."""
from pathlib import Path

import numpy as np
import torch

from panacea_amd import sampling as S

G = np.load(Path(__file__).resolve().parent / "golden" / "sampler.npz")


def fake_network(x, t, c):
    return torch.tanh(0.3 * x) * 0.5 + 1e-4 * t.float()[:, None, None, None] + 0.01 * c["crossattn"].mean() \
        + 0.05 * c["concat"]


def _inputs():
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(2, 4, 4, 12, generator=g)
    c = {"crossattn": torch.randn(1, 77, 8, generator=g), "concat": torch.randn(2, 4, 4, 12, generator=g),
         "cond_feat": torch.rand(2, 19, 8, 8, generator=g)}
    uc = {"crossattn": torch.randn(1, 77, 8, generator=g), "concat": c["concat"].clone(), "cond_feat": c["cond_feat"].clone()}
    return x0, c, uc


def test_discretization_and_denoiser_table():
    den = S.DiscreteDenoiser()
    assert np.allclose(den.sigmas.numpy(), G["denoiser.sigmas"], rtol=1e-6, atol=0)
    for n in (3, 25):
        assert np.allclose(S.LegacyDDPMDiscretization()(n).numpy(), G[f"sampler.{n}.sigmas"], rtol=1e-6)
    assert abs(float(S.LegacyDDPMDiscretization()(25)[0]) - 14.614641) < 1e-4
    assert S.timestep_indices(3) == [999, 666, 333]
    assert S.timestep_indices(25)[:2] == [999, 959] and S.timestep_indices(50)[:2] == [999, 979]


def test_euler_cfg_trajectory_matches_reference():
    x0, c, uc = _inputs()
    den = S.DiscreteDenoiser()
    for n in (3, 25):
        seen = []

        def net(x, t, cc):
            seen.append(t.clone())
            return fake_network(x, t, cc)
        smp = S.EulerEDMSampler(n, guider=S.VanillaCFG(5.0), device="cpu")
        xs = smp(lambda inp, sigma, cc: den(net, inp, sigma, cc), x0.clone(), c, uc)
        assert np.array_equal(torch.stack(seen)[:, 0].numpy(), G[f"sampler.{n}.timesteps"])
        assert seen[0].dtype == torch.int64 and seen[0].shape == (4,)         # CFG doubles the batch
        assert np.allclose(xs.numpy(), G[f"sampler.{n}.x_final"], atol=2e-5, rtol=1e-5)
