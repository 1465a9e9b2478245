"""Drop the MI355X path into an unmodified checkout of the reference.

The reference builds its network purely from YAML `target:` strings resolved by
`instantiate_from_config` -> `getattr(importlib.import_module(module), cls)` (sgm/util.py:168-185):

    sgm.modules.diffusionmodules.controlmodel.ControlledUNetModel3D   (inference_nuscenes.yaml:31)
    sgm.modules.diffusionmodules.controlmodel.ControlNet3D            (inference_nuscenes.yaml:52)
    sgm.modules.diffusionmodules.wrappers.OpenAIWrapperControlLDM3D   (diffusion.py:72, wrappers.py:6-8)

`install()` rebinds exactly these three attributes to the mirror classes of `panacea_amd.nn`, whose module
trees and state-dict keys equal the reference's — so `DiffusionEngine3D`, the sampler, the conditioner, the VAE,
the checkpoint loader (`load_state_dict(strict=False)`, inference.py:219) and `inference.py` itself run unchanged,
while every eps_theta evaluation goes through libpanacea_hip.so.  Nothing else of `sgm` is touched (the VAE and the
text encoder keep importing the reference's own attention / util modules).

Two ways to use it:
  * one line in a launcher:           import panacea_amd.dropin; panacea_amd.dropin.install()
  * zero lines (inference.py as is):  PYTHONPATH=<repo>/dropin_site:<repo>:<reference> python inference.py ...
    (`dropin_site/sitecustomize.py` calls `install(lazy=True)`: the rebinding happens right after the reference's
    modules are first imported.)
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import sys

TARGETS = {
    "sgm.modules.diffusionmodules.controlmodel": ("ControlNet3D", "ControlledUNetModel3D"),
    "sgm.modules.diffusionmodules.wrappers": ("OpenAIWrapperControlLDM3D",),
}
# opt-in (install(first_stage=True) / PANACEA_DROPIN_FIRST_STAGE=1): the first-stage Decoder.  `sgm/models/autoencoder.py`
# does `from ..modules.diffusionmodules.model import Decoder, Encoder`, so the class is rebound right after model.py
# is first imported; the Encoder, quant / post_quant convs and the engine stay the reference's.
FIRST_STAGE_TARGETS = {"sgm.modules.diffusionmodules.model": ("Decoder",)}
# opt-in (install(conditioner=True) / PANACEA_DROPIN_CONDITIONER=1): the conditioner classes the YAML names under
# `conditioner_config` (inference_nuscenes.yaml:72-96).  The text tower then runs on the kernels with the weights the
# engine checkpoint carries under `conditioner.embedders.0.model.*`; tokenisation stays open_clip's.
CONDITIONER_TARGETS = {"sgm.modules.encoders.modules": ("GeneralConditioner", "IdentityEncoder", "VAEEmbedder",
                                                        "FrozenOpenCLIPEmbedder")}


# Bindings of the rebound classes that OTHER modules of the reference hold by value (`from .encoders.modules import
# GeneralConditioner` in sgm/modules/__init__.py — and `sgm.modules.GeneralConditioner` is what the YAML names —;
# `from ..modules.encoders.modules import VAEEmbedder` in sgm/models/diffusion.py, used in an isinstance check).  When those
# modules are already imported at install() time their copies are rebound too; modules imported later pick the patched
# attribute up by themselves.
RE_EXPORTS = {"sgm.modules.encoders.modules": ("sgm.modules", "sgm.models.diffusion", "sgm.modules.encoders"),
              "sgm.modules.diffusionmodules.model": ("sgm.models.autoencoder",),
              "sgm.modules.diffusionmodules.wrappers": ("sgm.models.diffusion",)}


def _patch(module) -> None:
    from . import nn as mirror
    from .nn import model as first_stage
    if module.__name__.endswith("encoders.modules"):
        from . import conditioner as src
    else:
        src = first_stage if module.__name__.endswith("diffusionmodules.model") else mirror
    for cls in TARGETS[module.__name__]:
        ref_cls = getattr(module, cls, None)
        if ref_cls is getattr(src, cls):
            continue                                       # already patched
        setattr(module, "_reference_" + cls, ref_cls)
        setattr(module, cls, getattr(src, cls))
        for other in RE_EXPORTS.get(module.__name__, ()):
            om = sys.modules.get(other)
            if om is not None and ref_cls is not None and getattr(om, cls, None) is ref_cls:
                setattr(om, cls, getattr(src, cls))
    if module.__name__.endswith("controlmodel"):
        # the mirror ControlledUNetModel3D resolves `controlnet_config.target` through the same (patched) module
        module._panacea_amd = True


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        _patch(module)


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name not in TARGETS:
            return None
        for f in sys.meta_path:
            if f is self or not hasattr(f, "find_spec"):
                continue
            spec = f.find_spec(name, path, target)
            if spec is not None and spec.loader is not None:
                spec.loader = _PatchingLoader(spec.loader)
                return spec
        return None


_installed = False


def install(lazy: bool = False, first_stage: bool = False, conditioner: bool = False) -> None:
    """Rebind the three network classes (and, with first_stage=True, the VAE `Decoder`; with conditioner=True the
    conditioner classes).  lazy=False imports the reference modules now (they must be importable); lazy=True only arms
    an import hook."""
    global _installed
    if first_stage:
        TARGETS.update(FIRST_STAGE_TARGETS)
    if conditioner:
        TARGETS.update(CONDITIONER_TARGETS)
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        _installed = True
    for name in TARGETS:
        if name in sys.modules:
            _patch(sys.modules[name])
        elif not lazy:
            importlib.import_module(name)          # goes through _Finder -> patched on load
