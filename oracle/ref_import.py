"""Import the reference's hot-path modules from /root/reference WITHOUT running its package
`__init__`s (which need pytorch_lightning / open_clip / kornia, absent here).

Runs only in the build container (where /root/reference exists); used by `oracle/gen_golden.py`
to pin the oracle and to produce `tests/golden/`.  Holds no reference code: it only arranges
`sys.modules` (namespace shells for `sgm`, stubs for `omegaconf` and `xformers.ops`) and then
lets Python import the reference's own files from where they lie (recipe: SURVEY.md Appendix A).
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

import torch

REF = Path("/root/reference")


def available() -> bool:
    return (REF / "sgm" / "modules" / "attention.py").exists()


def import_reference():
    """Returns a namespace with the reference modules: .att .oai .cm .wr .dn .sp .util"""
    if not available():
        raise RuntimeError("/root/reference is not present (golden vectors are generated in the build container only)")

    def shell(name: str, path: Path):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m

    shell("sgm", REF / "sgm")
    shell("sgm.modules", REF / "sgm" / "modules")
    shell("sgm.modules.diffusionmodules", REF / "sgm" / "modules" / "diffusionmodules")

    oc = types.ModuleType("omegaconf")

    class ListConfig(list):
        pass

    class OmegaConf(dict):
        pass

    oc.ListConfig, oc.OmegaConf = ListConfig, OmegaConf
    lc = types.ModuleType("omegaconf.listconfig")
    lc.ListConfig = ListConfig
    sys.modules["omegaconf"], sys.modules["omegaconf.listconfig"] = oc, lc

    # xformers 0.0.16 memory_efficient_attention on (B*heads, N, d): softmax(q k^T d^-1/2) v
    xf, xo = types.ModuleType("xformers"), types.ModuleType("xformers.ops")
    xo.memory_efficient_attention = lambda q, k, v, attn_bias=None, op=None: \
        torch.nn.functional.scaled_dot_product_attention(q, k, v)
    xf.ops = xo
    sys.modules["xformers"], sys.modules["xformers.ops"] = xf, xo

    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    ns = types.SimpleNamespace()
    ns.util = importlib.import_module("sgm.util")
    ns.att = importlib.import_module("sgm.modules.attention")
    ns.oai = importlib.import_module("sgm.modules.diffusionmodules.openaimodel")
    ns.cm = importlib.import_module("sgm.modules.diffusionmodules.controlmodel")
    ns.wr = importlib.import_module("sgm.modules.diffusionmodules.wrappers")
    ns.dn = importlib.import_module("sgm.modules.diffusionmodules.denoiser")
    ns.sp = importlib.import_module("sgm.modules.diffusionmodules.sampling")
    ns.dutil = importlib.import_module("sgm.modules.diffusionmodules.util")
    return ns


def build_reference_network(ns, unet_kwargs: dict, hint_channels: int = 19, control_scales: float = 1.0,
                            with_controlnet: bool = True, device=None):
    """ControlledUNetModel3D(+ControlNet3D) wrapped like DiffusionEngine3D does (diffusion.py:71-78)."""
    import contextlib
    import io
    cn_cfg = None
    if with_controlnet:
        cn_cfg = {"target": "sgm.modules.diffusionmodules.controlmodel.ControlNet3D",
                  "params": dict(unet_kwargs, hint_channels=hint_channels, control_scales=control_scales)}
    ctx = torch.device(device) if device is not None else contextlib.nullcontext()
    with contextlib.redirect_stdout(io.StringIO()), ctx:
        net = ns.cm.ControlledUNetModel3D(controlnet_config=cn_cfg, out_channels=4, **unet_kwargs).eval()
        wrapper = ns.wr.OpenAIWrapperControlLDM3D(net)
    return net, wrapper
