"""panacea_amd — MI355X-native (gfx950) implementation of Panacea's denoising hot path.

`panacea_amd.nn` mirrors the reference's module API (sgm.modules.diffusionmodules.{wrappers,controlmodel,
openaimodel}, sgm.modules.attention); `panacea_amd.hip` binds the hand-written HIP kernels in
`panacea_amd/csrc` through the C-ABI of `include/panacea_hip.h`.  See DESIGN.md.
"""
__version__ = "0.1.0"


def build_network(cfg: dict, hint_channels: int = 19, control_scales: float = 1.0):
    """ControlledUNetModel3D (+ ControlNet3D) wrapped like DiffusionEngine3D does (diffusion.py:71-78)."""
    from .nn import ControlledUNetModel3D, OpenAIWrapperControlLDM3D
    cn = {"target": "panacea_amd.nn.controlmodel.ControlNet3D",
          "params": dict(cfg, hint_channels=hint_channels, control_scales=control_scales)}
    cn["params"].pop("out_channels", None)
    net = ControlledUNetModel3D(controlnet_config=cn, out_channels=4, **cfg).eval()
    return OpenAIWrapperControlLDM3D(net)
