"""Golden vectors of the conditioner ROUTING: runs the reference's own `GeneralConditioner`
(sgm/modules/encoders/modules.py:95-220, imported from /root/reference in the build container) over toy embedders and a
synthetic nuScenes-shaped batch, and stores inputs + the c / uc dictionaries in tests/golden/conditioner.npz.

    python -m oracle.gen_golden_conditioner

The toy embedders subclass the REFERENCE's AbstractEmbModel; tests/test_conditioner.py defines the same three toys on the
mirror's AbstractEmbModel.  No reference code is stored: data only.
"""
import importlib
import sys
import types
from pathlib import Path

import numpy as np
import torch

from oracle import ref_import as R

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"


def import_conditioner():
    R.import_reference()

    def shell(name, path):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m
    for n, p in [("sgm.modules.encoders", "sgm/modules/encoders"), ("sgm.modules.autoencoding", "sgm/modules/autoencoding"),
                 ("sgm.modules.distributions", "sgm/modules/distributions")]:
        if n not in sys.modules:
            shell(n, R.REF / p)
    for stub in ("kornia", "open_clip"):            # imported at module level by the reference, not used by the routing
        sys.modules.setdefault(stub, types.ModuleType(stub))
    return importlib.import_module("sgm.modules.encoders.modules")


def batch(seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"txt": torch.randn(2, 77, 16, generator=g),                      # stands in for the tokenised prompt
            "cond_img": torch.rand(2, 3, 19, 16, 24, generator=g),           # (b, t, c, h, w) BEV layout
            "final_cond_zero": torch.randn(2, 3, 3, 16, 24, generator=g),    # (b, t, c, h, w) conditioning frames
            "size": torch.randn(2, 6, generator=g)}


def main():
    m = import_conditioner()
    toys = types.ModuleType("oracle_toy_embedders")

    class ToyText(m.AbstractEmbModel):          # rank-3 output -> crossattn
        def forward(self, x):
            return torch.tanh(x) * 2.0

    class ToyVAE(m.AbstractEmbModel):           # rank-4 output -> concat
        def forward(self, x):
            return torch.nn.functional.avg_pool2d(x, 8)[:, :, :, :] * 0.18215

    class ToyVector(m.AbstractEmbModel):        # rank-2 output -> vector
        def forward(self, x):
            return x * 3.0
    toys.ToyText, toys.ToyVAE, toys.ToyVector = ToyText, ToyVAE, ToyVector
    sys.modules["oracle_toy_embedders"] = toys
    cfg = [{"target": "oracle_toy_embedders.ToyText", "input_key": "txt", "ucg_rate": 0.1, "is_trainable": False},
           {"target": "sgm.modules.encoders.modules.IdentityEncoder", "input_key": "cond_img", "ucg_rate": 0.0},
           {"target": "oracle_toy_embedders.ToyVAE", "input_key": "final_cond_zero", "ucg_rate": 0.0},
           {"target": "oracle_toy_embedders.ToyVector", "input_key": "size", "ucg_rate": 0.0},
           {"target": "oracle_toy_embedders.ToyVector", "input_key": "size", "ucg_rate": 0.0}]     # second vector: concatenated
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        cond = m.GeneralConditioner(cfg)
    b = batch()
    c, uc = cond.get_unconditional_conditioning({k: v.clone() for k, v in b.items()},
                                                force_uc_zero_embeddings=["txt"])
    out = {f"in.{k}": v.numpy() for k, v in b.items()}
    out.update({f"c.{k}": v.numpy() for k, v in c.items()})
    out.update({f"uc.{k}": v.numpy() for k, v in uc.items()})
    np.savez_compressed(GOLDEN / "conditioner.npz", **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
