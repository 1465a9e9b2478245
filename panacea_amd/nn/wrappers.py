"""Network wrapper — host-side mirror of `sgm/modules/diffusionmodules/wrappers.py` (the drop-in
boundary: `Denoiser.__call__` -> `network(x * c_in, c_noise, cond)`, denoiser.py:28)."""
from __future__ import annotations

import torch
import torch.nn as nn

OPENAIUNETWRAPPER = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapper"
OPENAIUNETWRAPPERCONTROLLDM3D = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapperControlLDM3D"


class IdentityWrapper(nn.Module):
    """wrappers.py:10-22.  `compile_model` is accepted and ignored: the path already consists of
    hand-written kernels and is hipGraph-capturable (panacea_amd.graph), there is nothing to trace."""

    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    """wrappers.py:25-36"""

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        if "concat" in c:
            x = torch.cat((x, c["concat"]), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)


class OpenAIWrapperControlLDM3D(IdentityWrapper):
    """wrappers.py:37-70: eps = UNet(cat(x, concat), t, text, control=ControlNet(cat(x, concat), BEV hint, t, text))."""

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        net = self.diffusion_model
        model_dtype = net.controlnet.input_hint_block[0].weight.dtype
        if "concat" in c:
            x = torch.cat((x, c["concat"].to(x.dtype)), dim=1)
        # the reference casts the caller's dict entry in place (wrappers.py:48); keep that side effect
        c["crossattn"] = c["crossattn"].to(model_dtype)
        if c.get("vector", None) is not None:
            raise NotImplementedError("class-conditional `vector` conditioning is not on the Panacea path")
        # "_invariants": optional ControlledUNetModel3D.prepare() result placed in `c` by a sampler that keeps the
        # conditioning fixed over its steps (panacea_amd.sampling); never present when the reference's sampler calls
        out = net.denoise(x, t, c["crossattn"], c["cond_feat"], trace=kwargs.get("trace"),
                          invariants=c.get("_invariants"))
        return out.to(model_dtype)
