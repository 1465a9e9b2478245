"""Checkpoint formats and image writers at the edges of the path: the three on-disk formats the
reference's `model_load_ckpt` accepts load into the mirror network with the reference's key names."""
import torch

import emu
from helpers import cond, product_network, step_inputs
from panacea_amd import checkpoint as ck, engine as E


def test_three_checkpoint_formats_load_and_change_the_output(tmp_path):
    w, sd, kw = product_network("tiny")
    other, sd2, _ = product_network("tiny", salt=3)                   # a second set of weights = "the checkpoint"
    engine_sd = {ck.DENOISER_PREFIX + k: v for k, v in sd2.items()}
    engine_sd["first_stage_model.decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)     # ignored sub-trees
    engine_sd["conditioner.embedders.0.x"] = torch.zeros(3)
    files = {}
    torch.save({"state_dict": engine_sd}, tmp_path / "last.ckpt"); files["lightning"] = tmp_path / "last.ckpt"
    torch.save({"_forward_module." + k: v for k, v in engine_sd.items()}, tmp_path / "deepspeed_model.ckpt")
    files["deepspeed"] = tmp_path / "deepspeed_model.ckpt"
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in engine_sd.items()}, str(tmp_path / "model.safetensors"))
    files["safetensors"] = tmp_path / "model.safetensors"
    inp = step_inputs("tiny", kw)
    with E.use_backend(emu):
        want = other(inp["x"], inp["t"], cond(inp))
        before = w(inp["x"], inp["t"], cond(inp))
        assert not torch.equal(before, want)
        for name, path in files.items():
            w.diffusion_model.load_state_dict(sd, strict=True)        # back to the first weights
            missing, unexpected = ck.load_denoiser(w, str(path), verbose=False)
            assert missing == [] and unexpected == [], (name, missing[:3], unexpected[:3])
            assert torch.equal(w(inp["x"], inp["t"], cond(inp)), want), name     # packed fp16 copies were refreshed
    # a partial checkpoint reports what is missing instead of failing (strict=False, inference.py:216)
    part = {k: v for k, v in engine_sd.items() if "controlnet" not in k}
    torch.save({"state_dict": part}, tmp_path / "part.ckpt")
    missing, unexpected = ck.load_denoiser(w, str(tmp_path / "part.ckpt"), verbose=False)
    assert missing and all(k.startswith("controlnet.") for k in missing) and unexpected == []


def test_image_writers(tmp_path):
    from PIL import Image
    T, H, Wv = 3, 8, 16
    frames = torch.linspace(-1.2, 1.2, T * 3 * H * 6 * Wv).view(T, 3, H, 6 * Wv)
    names = [f"clip_cam{i}" for i in range(6)]
    paths = ck.save_view_frames(frames, str(tmp_path), names, view_width=Wv)
    assert len(paths) == 6 * T
    im = Image.open(tmp_path / "clip_cam3" / "_000001.jpg")
    assert im.size == (Wv, H) and im.mode == "RGB"
    g = Image.open(ck.save_gif(frames, str(tmp_path / "pano.gif")))
    assert g.n_frames == T and g.size == (6 * Wv, H) and g.info.get("loop") == 0


def test_lightning_checkpoint_with_hyper_parameters_needs_trust(tmp_path):
    """a Lightning .ckpt that carries non-tensor objects is refused by the safe unpickler and loads with
    trust=True (the reference always unpickles fully, inference.py:209)."""
    import pickle
    import types
    import pytest
    import torch
    from panacea_amd import checkpoint as ck
    hp = types.SimpleNamespace(lr=1e-4)                         # stands in for an OmegaConf container
    path = str(tmp_path / "lightning.ckpt")
    torch.save({"state_dict": {"model.diffusion_model.a": torch.ones(3)}, "hyper_parameters": hp, "epoch": 3}, path)
    with pytest.raises(pickle.UnpicklingError):
        ck.read_state_dict(path)
    sd = ck.read_state_dict(path, trust=True)
    assert list(sd) == ["model.diffusion_model.a"] and torch.equal(sd["model.diffusion_model.a"], torch.ones(3))
