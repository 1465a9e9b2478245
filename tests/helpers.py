"""Shared test helpers: build the product network with deterministic synthetic weights, load golden
vectors, compare with the oracle."""
import contextlib
import json
from pathlib import Path

import numpy as np
import torch

from oracle import panacea_oracle as po
from panacea_amd import build_network, configs, synth

GOLDEN = Path(__file__).resolve().parent / "golden"


def oracle_cfg(kw, **over):
    c = po.OracleConfig(num_frames=kw["num_frames"], model_channels=kw["model_channels"],
                        num_head_channels=kw["num_head_channels"],
                        spatial_only_attn_type=kw["spatial_only_attn_type"],
                        insert_crossview=kw["insert_crossview"])
    for k, v in over.items():
        setattr(c, k, v)
    return c


def manifest(name):
    return json.loads((GOLDEN / f"manifest_{name}.json").read_text())


def golden(name):
    return np.load(GOLDEN / f"{name}.npz")


def product_network(name, device="cpu", kw=None, salt=0):
    kw = kw or configs.get(name)
    w = build_network(kw)
    sd = synth.synth_state_dict(manifest(name), salt=salt)
    w.diffusion_model.load_state_dict(sd, strict=True)
    return w.to(device), sd, kw


def step_inputs(name, kw, device="cpu", t_index=999, shape=None):
    B, T, h, w = shape or configs.SHAPES[name]
    inp = synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"], t_index=t_index)
    return {k: v.to(device) for k, v in inp.items()}


def cond(inp):
    return {k: inp[k] for k in ("concat", "crossattn", "cond_feat")}


def err_stats(got: torch.Tensor, ref) -> dict:
    ref = torch.as_tensor(ref).float()
    d = (got.detach().float().cpu() - ref).abs()
    return dict(max_abs=d.max().item(), mean_abs=d.mean().item(), ref_rms=ref.pow(2).mean().sqrt().item(),
                ref_max=ref.abs().max().item())


@contextlib.contextmanager
def gn_statistics_from_launches():
    """Every spatial GroupNorm with its own statistics launch (engine.GN_FROM_EPILOGUE off): the form in which a run that only MOVES
    data (round 2's transposed frame shard) must reproduce the unsharded bits — by default the unsharded ResBlock3D takes two of
    its GroupNorms' statistics from the temporal convs' epilogues, the transposed one cannot (other fp32 summation order)."""
    from panacea_amd import engine as E
    prev = E.GN_FROM_EPILOGUE
    E.GN_FROM_EPILOGUE = False
    try:
        yield
    finally:
        E.GN_FROM_EPILOGUE = prev


def measured(tag: str, **values):
    """append one line of measured numbers to gpurun_out/test_measurements.log (GPU runs: the gates next to each test are
    pinned from these; no-op when the directory does not exist)"""
    d = Path(__file__).resolve().parent.parent / "gpurun_out"
    if d.is_dir():
        with open(d / "test_measurements.log", "a") as f:
            f.write(tag + " " + " ".join(f"{k}={v:.4e}" if isinstance(v, float) else f"{k}={v}" for k, v in values.items()) + "\n")
