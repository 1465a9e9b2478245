"""Generate tests/golden/vae_tiny.npz by running the REFERENCE's own `Decoder` (sgm/modules/diffusionmodules/model.py)
in the build container.  Data only goes into the repo: the synthetic-weight manifest (names + shapes), the input z,
the reference's output image and two intermediate maps.  Usage: python oracle/gen_golden_vae.py"""
import contextlib
import importlib
import io
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref_import, vae_oracle as vo       # noqa: E402
from panacea_amd import synth                         # noqa: E402

TINY = dict(double_z=True, z_channels=4, resolution=16, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1,
            attn_resolutions=[], dropout=0.0)


def main():
    ref_import.import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        mdl = importlib.import_module("sgm.modules.diffusionmodules.model")
        dec = mdl.Decoder(**TINY).eval()
    pq = torch.nn.Conv2d(4, 4, 1)                      # AutoencoderKL.post_quant_conv (autoencoder.py:349)
    names = {"decoder." + k: list(v.shape) for k, v in dec.state_dict().items()}
    names.update({"post_quant_conv." + k: list(v.shape) for k, v in pq.state_dict().items()})
    (ROOT / "tests" / "golden" / "manifest_vae_tiny.json").write_text(json.dumps(names, indent=0))
    sd = synth.synth_state_dict(names)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=True)
    pq.load_state_dict({k[len("post_quant_conv."):]: v for k, v in sd.items() if k.startswith("post_quant_conv.")})
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 4, 8, 48, generator=g) * 2.0
    with torch.no_grad():
        img = dec(pq(z))
        tr = {}
        ora = vo.decode(sd, vo.VaeConfig(ch=64, ch_mult=[1, 2], num_res_blocks=1), z, trace=tr)
    err = (img - ora).abs().max().item()
    print(f"reference decoder {tuple(img.shape)}  |img| max {img.abs().max():.3f} rms {img.pow(2).mean().sqrt():.3f}; "
          f"oracle vs reference max-abs {err:.2e}")
    assert err < 1e-4
    np.savez_compressed(ROOT / "tests" / "golden" / "vae_tiny.npz", z=z.numpy(), img=img.numpy(),
                        mid_s7=tr["mid"].reshape(-1)[::7].numpy(), up1_s7=tr["up.1"].reshape(-1)[::7].numpy(),
                        oracle_vs_reference=np.float32(err))
    # ---- encoder: the reference's own Encoder + a quant_conv (autoencoder.py:346,348)
    with contextlib.redirect_stdout(io.StringIO()):
        enc = mdl.Encoder(**TINY).eval()
    qc = torch.nn.Conv2d(8, 8, 1)
    enames = {"encoder." + k: list(v.shape) for k, v in enc.state_dict().items()}
    enames.update({"quant_conv." + k: list(v.shape) for k, v in qc.state_dict().items()})
    (ROOT / "tests" / "golden" / "manifest_vae_enc_tiny.json").write_text(json.dumps(enames, indent=0))
    esd = synth.synth_state_dict(enames)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in esd.items() if k.startswith("encoder.")}, strict=True)
    qc.load_state_dict({k[len("quant_conv."):]: v for k, v in esd.items() if k.startswith("quant_conv.")})
    x = torch.tanh(torch.randn(2, 3, 16, 96, generator=g))
    with torch.no_grad():
        mom = qc(enc(x))
        etr = {}
        eora = vo.encode_moments(esd, vo.VaeConfig(ch=64, ch_mult=[1, 2], num_res_blocks=1), x, trace=etr)
    eerr = (mom - eora).abs().max().item()
    print(f"reference encoder {tuple(mom.shape)}  |moments| max {mom.abs().max():.3f} rms {mom.pow(2).mean().sqrt():.3f}; "
          f"oracle vs reference max-abs {eerr:.2e}")
    assert eerr < 1e-4
    np.savez_compressed(ROOT / "tests" / "golden" / "vae_enc_tiny.npz", x=x.numpy(), moments=mom.numpy(),
                        down0_s7=etr["down.0"].reshape(-1)[::7].numpy(), oracle_vs_reference=np.float32(eerr))


if __name__ == "__main__":
    main()
