"""The callers either side of the hot path: discretisation, discrete denoiser with
eps-scaling, classifier-free guidance and the Euler sampler of `configs/inference_nuscenes.yaml`.

Host-side mirrors with the reference's names and call signatures
  LegacyDDPMDiscretization   sgm/modules/diffusionmodules/discretizer.py:42-69
  EpsScaling                 .../denoiser_scaling.py:16-22
  DiscreteDenoiser           .../denoiser.py:31-63
  VanillaCFG                 .../guiders.py:8-40 (+ sampling_utils.py:7-9)
  EulerEDMSampler            .../sampling.py:27-133,214-218
written for a device-resident loop: the sigma schedule is a Python list of floats plus one device tensor
(no `.item()` sync per step — the reference compares Python floats with a device element at
sampling.py:118-122), and every per-step quantity (c_in, c_out, timestep index) is a host scalar.
These are a handful of elementwise torch ops per step; the arithmetic that matters is in the network.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    return x[(...,) + (None,) * (target_dims - x.ndim)]


class LegacyDDPMDiscretization:
    """discretizer.py:42-69 — sigma_i = sqrt((1 - abar_i) / abar_i) of the 1000-step linear-beta DDPM schedule."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=np.float64) ** 2
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            ts = np.linspace(self.num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
            ac = self.alphas_cumprod[ts]
        elif n == self.num_timesteps:
            ac = self.alphas_cumprod
        else:
            raise ValueError("more sampling steps than training timesteps")
        sig = torch.tensor((1 - ac) / ac, dtype=torch.float32, device=device) ** 0.5
        return torch.flip(sig, (0,))

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        s = self.get_sigmas(n, device=device)
        if do_append_zero:
            s = torch.cat([s, s.new_zeros([1])])
        return s if not flip else torch.flip(s, (0,))


class EpsScaling:
    """denoiser_scaling.py:16-22"""

    def __call__(self, sigma):
        return torch.ones_like(sigma), -sigma, 1 / (sigma ** 2 + 1.0) ** 0.5, sigma.clone()


class DiscreteDenoiser(nn.Module):
    """denoiser.py:31-63 with EpsScaling: snaps sigma to the 1000-entry table and hands the network its INDEX."""

    def __init__(self, num_idx=1000, discretization=None, do_append_zero=False, quantize_c_noise=True, flip=True):
        super().__init__()
        disc = discretization or LegacyDDPMDiscretization()
        self.register_buffer("sigmas", disc(num_idx, do_append_zero=do_append_zero, flip=flip))
        self.scaling = EpsScaling()
        self.quantize_c_noise = quantize_c_noise

    def sigma_to_idx(self, sigma):
        return (sigma - self.sigmas[:, None]).abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self.sigmas[idx]

    def __call__(self, network: Callable, input: torch.Tensor, sigma: torch.Tensor, cond: Dict) -> torch.Tensor:
        sigma = self.idx_to_sigma(self.sigma_to_idx(sigma))
        shape = sigma.shape
        sigma = append_dims(sigma, input.ndim)
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = c_noise.reshape(shape)
        if self.quantize_c_noise:
            c_noise = self.sigma_to_idx(c_noise)
        return network(input * c_in, c_noise, cond) * c_out + input * c_skip


class VanillaCFG:
    """guiders.py:8-40: batch-doubling (uncond half first) and x_u + s (x_c - x_u)."""
    KEYS = ("vector", "crossattn", "concat", "cond_feat", "cond_bev_feat")

    def __init__(self, scale, dyn_thresh_config=None):
        self.scale = scale

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return x_u + self.scale * (x_c - x_u)

    def prepare_inputs(self, x, s, c, uc):
        c_out = {}
        pre = c.get("_cat")                       # hoist_invariants(): uc|c already concatenated once per schedule
        for k in c:
            if k == "_cat":
                continue
            if k in self.KEYS:
                c_out[k] = pre[k] if pre is not None else torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] is uc[k] or c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


def _same_tensor(a: torch.Tensor, b: torch.Tensor) -> bool:
    """the two names refer to the same values in the same memory (identity of content without reading it)"""
    return a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
                      and a.device == b.device)


class BoundDenoiser:
    """`lambda input, sigma, c: denoiser(model, input, sigma, c)` of DiffusionEngine3D.sample (diffusion.py:251-253) as an
    object: the sampler can see which denoiser and which network it drives and run the whole step on the device — the
    c_in scaling and the CFG batch doubling folded into the network's entry kernel, c_out / c_skip + CFG combine + Euler
    update in ONE exit kernel on the network's channels-last output.  Calling it is exactly the lambda."""

    def __init__(self, denoiser: "DiscreteDenoiser", network):
        self.denoiser, self.network = denoiser, network

    def __call__(self, x, sigma, cond):
        return self.denoiser(self.network, x, sigma, cond)


class EulerEDMSampler:
    """sampling.py:27-133,214-218 with s_churn = 0 (deterministic; == DDIM for eps-prediction)."""
    fuse = True          # use the fused device step when the denoiser is a BoundDenoiser around the HIP network

    def __init__(self, num_steps: int, guider: Optional[VanillaCFG] = None, discretization=None, device="cuda"):
        self.num_steps = num_steps
        self.discretization = discretization or LegacyDDPMDiscretization()
        self.guider = guider
        self.device = device

    def sigmas(self, num_steps=None) -> torch.Tensor:
        return self.discretization(self.num_steps if num_steps is None else num_steps, device=self.device)

    def denoise(self, x, denoiser, sigma, cond, uc):
        if self.guider is None:
            return denoiser(x, sigma, cond)
        return self.guider(denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc)), sigma)

    def _fusable(self, denoiser, x, cond) -> bool:
        if not (self.fuse and isinstance(denoiser, BoundDenoiser) and x.is_cuda):
            return False
        model = getattr(denoiser.network, "diffusion_model", None)
        den = denoiser.denoiser
        # frame- / view-sharded networks fuse as well (round 4): the entry and exit kernels are elementwise over whatever frames /
        # band the rank holds; a CFG pair (parallel.ShardedCFG: `half`, `group`) all-gathers its eps halves in front of the
        # exit kernel instead of the denoised halves behind it
        guider_ok = type(self.guider) in (VanillaCFG, type(None)) or (isinstance(self.guider, VanillaCFG) and hasattr(self.guider, "half"))
        return (hasattr(model, "denoise_tokens") and guider_ok and isinstance(den, DiscreteDenoiser)
                and isinstance(den.scaling, EpsScaling) and den.quantize_c_noise
                and "concat" in cond and cond.get("vector") is None)

    def _fused_step(self, sigma, next_sigma, denoiser, x, cond, uc):
        """One step with three tiny torch ops (table snap of T sigmas) + the network + one exit kernel; same arithmetic, in
        the reference's rounding order, as denoise() + the Euler update below."""
        from . import engine as E
        den, model = denoiser.denoiser, denoiser.network.diffusion_model
        T = x.shape[0]
        idx = den.sigma_to_idx(sigma)
        sig_q = den.idx_to_sigma(idx)                                 # denoiser.py:24
        c_in = 1 / (sig_q ** 2 + 1.0) ** 0.5
        c_noise = den.sigma_to_idx(sig_q)                             # quantised c_noise = the table index
        half = getattr(self.guider, "half", None)
        if self.guider is None:
            cat, inv, nh = cond, cond.get("_invariants"), 1
        elif half is not None:
            # one CFG half per rank: this rank evaluates its half; the pair's eps tokens are gathered for the exit kernel
            cat, inv, nh = (uc if half == 0 else cond), cond.get("_invariants"), 1
        else:
            pre = cond.get("_cat")
            if pre is not None:
                cat = pre
            else:
                cat = {k: torch.cat((uc[k], cond[k]), 0) for k in ("crossattn", "concat")}
                # the BEV-layout hint: uc and c normally hold the SAME tensor (IdentityEncoder returns its input for both
                # conditioner passes, modules.py:242-247).  Then it is handed over once — no 2 x 0.5 GB concatenation per step
                # and the network runs its hint stem on T frames instead of 2 T
                hu, hc = uc["cond_feat"], cond["cond_feat"]
                cat["cond_feat"] = hc if _same_tensor(hu, hc) else torch.cat((hu, hc), 0)
            inv, nh = cond.get("_invariants"), 2
        ctx = cat["crossattn"].to(model.controlnet.input_hint_block[0].weight.dtype)
        eps = model.denoise_tokens(x, c_in.repeat(nh).contiguous(), c_noise.repeat(nh).contiguous(), ctx, cat["concat"],
                                   cat["cond_feat"], invariants=inv)
        x32 = x.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x32)
        eps32, cfg = eps.f32, nh == 2
        if half is not None:
            import torch.distributed as dist
            mine = eps32.contiguous()
            stage = dist.get_backend(self.guider.group) == "gloo" and mine.is_cuda       # two processes on one GPU (tests)
            send = mine.cpu() if stage else mine
            both = torch.empty((2 * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)   # [uncond half; cond half]
            dist.all_gather_into_tensor(both, send, group=self.guider.group)
            eps32, cfg = both.to(mine.device), True
        E.backend().cfg_euler_step(eps32, eps.C, T, eps.N, x.shape[1], cfg, float(self.guider.scale) if cfg else 0.0,
                                   x32, (-sig_q).contiguous(), sigma.contiguous(), next_sigma.contiguous(), out)
        return out.to(x.dtype)

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None):
        if self._fusable(denoiser, x, cond):
            return self._fused_step(sigma, next_sigma, denoiser, x, cond, uc)
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        d = (x - denoised) / append_dims(sigma, x.ndim)
        return x + append_dims(next_sigma - sigma, x.ndim) * d

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, network=None):
        """`network`: optional OpenAIWrapperControlLDM3D.  When given, the step invariants of the conditioning (text
        K/V of every cross-attention site, ControlNet hint stem) are computed once for the whole schedule instead of
        once per step; the trajectory is bit-identical to the plain loop."""
        sig = self.sigmas(num_steps)
        uc = cond if uc is None else uc
        if network is not None:
            cond, uc = hoist_invariants(network, self.guider, cond, uc)
        x = x * torch.sqrt(1.0 + sig[0] ** 2.0)
        s_in = x.new_ones([x.shape[0]])
        for i in range(len(sig) - 1):
            x = self.sampler_step(s_in * sig[i], s_in * sig[i + 1], denoiser, x, cond, uc)
        return x


def share_noise_init(randn: torch.Tensor, concat: torch.Tensor, share_noise_level: float) -> torch.Tensor:
    """DiffusionEngine3D.sample (sgm/models/diffusion.py:242-249): the initial latent of every frame of a clip carries
    `share_noise_level` x the conditioning latent of the LAST frame (`concat[-1]` tiled over the frames)."""
    if share_noise_level <= 0.0:
        return randn
    return randn + concat[-1].unsqueeze(0).expand(randn.shape[0], *concat.shape[1:]) * share_noise_level


def hoist_invariants(network, guider, cond: Dict, uc: Dict):
    """Returns copies of (cond, uc) that carry the network's StepInvariants for the batch the guider will build from
    them.  The concatenated conditioning tensors are built ONCE here and shared by every step (prepare_inputs would
    otherwise re-concatenate them per step, which also changes their identity)."""
    model = network.diffusion_model
    half = getattr(guider, "half", None)
    if half is not None:
        # parallel.ShardedCFG: this rank evaluates ONE CFG half with that half's own tensors (same objects every step)
        src = dict(uc if half == 0 else cond)
        src["crossattn"] = src["crossattn"].to(model.controlnet.input_hint_block[0].weight.dtype)
        inv = model.prepare(src["crossattn"], src["cond_feat"])
        c2, u2 = dict(cond), dict(uc)
        (u2 if half == 0 else c2).update(crossattn=src["crossattn"])
        c2["_invariants"] = u2["_invariants"] = inv
        return c2, u2
    if guider is None:
        c2 = dict(cond)
        c2["crossattn"] = cond["crossattn"].to(model.controlnet.input_hint_block[0].weight.dtype)
        c2["_invariants"] = model.prepare(c2["crossattn"], c2["cond_feat"])
        return c2, c2
    cat = {k: torch.cat((uc[k], cond[k]), 0) for k in cond if k in guider.KEYS}
    cat["crossattn"] = cat["crossattn"].to(model.controlnet.input_hint_block[0].weight.dtype)
    inv = model.prepare(cat["crossattn"], cat["cond_feat"])
    c2, u2 = dict(cond), dict(uc)
    c2["_invariants"] = u2["_invariants"] = inv
    c2["_cat"] = u2["_cat"] = cat
    return c2, u2


def timestep_indices(num_steps: int) -> List[int]:
    """The int64 timestep indices the network sees over a `num_steps` schedule (999, 959, ... for 25)."""
    den = DiscreteDenoiser()
    sig = LegacyDDPMDiscretization()(num_steps)[:-1]
    return den.sigma_to_idx(sig).tolist()
