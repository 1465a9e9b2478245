"""Module-level parity of the HIP path against the ORACLE's own functions (oracle/panacea_oracle.py — the restatement that is
pinned on the reference), not against the torch emulation of the C-ABI: ResBlock3D and SpatialTemporalTransformer at the
network's real widths, through their reference-compatible NCHW entry points.  (VERDICT r1 weak spot 3: kernel tests used
tests/emu.py only; a shared misunderstanding between emu and a kernel would have been caught at stride-7 block granularity
at best.)"""
import pytest
import torch

from helpers import measured
from oracle import panacea_oracle as po
from panacea_amd import synth
from panacea_amd.nn import attention as A, openaimodel as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


# (max-abs, mean-abs) of the module outputs (|out| ~ 5-7), pinned ~1.5x above the MI355X measurements of round 3
# (gpurun_out/test_measurements.log).  `precise` (the modules' default): ResBlock3D 4.8-6.3e-4 / 7.2-8.6e-5, STT 8.0-8.8e-4 /
# 1.3-1.5e-4;  `fast` (plain fp16 operands, what round 2's entries ran): 0.7-1.3e-3 / 0.9-1.9e-4 and 1.7-1.8e-3 / 2.9-3.1e-4
RES_TOL, RES_TOL_FAST = (1.0e-3, 1.4e-4), (2.0e-3, 3.0e-4)
STT_TOL, STT_TOL_FAST = (1.35e-3, 2.3e-4), (2.7e-3, 4.6e-4)


def _load(module, prefix, salt=0):
    man = {f"{prefix}.{k}": list(v.shape) for k, v in module.state_dict().items()}
    sd = synth.synth_state_dict(man, salt=salt, threads=1)
    module.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()}, strict=True)
    return sd


def _cfg(T, crossview=True):
    return po.OracleConfig(num_frames=T, model_channels=320, num_head_channels=64, spatial_only_attn_type="intra-view",
                           insert_crossview=crossview)


@pytest.mark.parametrize("cin,cout,T", [(320, 320, 2), (960, 640, 4), (640, 1280, 1)])
def test_resblock3d_vs_oracle(cin, cout, T):
    blk = O.ResBlock3D(cin, 1280, 0.0, out_channels=cout, dims=2, num_frames=T).eval()
    sd = _load(blk, "blk", salt=cin)
    g = torch.Generator().manual_seed(cin + T)
    x = torch.randn(2 * T, cin, 8, 96, generator=g)
    emb = torch.randn(2 * T, 1280, generator=g)
    ref = po.resblock3d(sd, "blk", x, emb, _cfg(T))
    got = blk.to(DEV)(x.to(DEV), emb.to(DEV)).cpu()
    err = (got - ref).abs()
    print(f"ResBlock3D {cin}->{cout} T={T}: max {err.max().item():.3e} mean {err.mean().item():.3e} |ref| {ref.abs().max().item():.2f}")
    assert got.shape == ref.shape
    assert blk.precision == "precise"          # the module entry runs the product's default operand policy (round 2: engine.FAST)
    measured("resblock3d_vs_oracle", cin=cin, cout=cout, T=T, max_abs=err.max().item(), mean_abs=err.mean().item(),
             ref_max=ref.abs().max().item())
    assert err.max().item() <= RES_TOL[0] and err.mean().item() <= RES_TOL[1]
    blk.precision = "fast"                     # plain fp16 operands; emulation of the same graph: 0.6-1.3e-3 / 0.9-1.9e-4
    errf = (blk(x.to(DEV), emb.to(DEV)).cpu() - ref).abs()
    measured("resblock3d_vs_oracle_fast", cin=cin, max_abs=errf.max().item(), mean_abs=errf.mean().item())
    assert errf.max().item() <= RES_TOL_FAST[0] and errf.mean().item() <= RES_TOL_FAST[1]


@pytest.mark.parametrize("C,T,h,w", [(320, 2, 8, 96), (640, 4, 4, 48), (1280, 2, 2, 24)])
def test_spatial_temporal_transformer_vs_oracle(C, T, h, w):
    stt = A.SpatialTemporalTransformer(C, C // 64, 64, depth=1, context_dim=1024, use_linear=True, num_frames=T,
                                       spatial_only_attn_type="intra-view", insert_crossview=True).eval()
    sd = _load(stt, "stt", salt=C)
    g = torch.Generator().manual_seed(C + T)
    B = 2
    x = torch.randn(B * T, C, h, w, generator=g)
    ctx = torch.randn(B, 77, 1024, generator=g)
    ctx_t = ctx[:, None].expand(B, T, 77, 1024).reshape(B * T, 77, 1024).contiguous()      # tiled over T like controlmodel.py:121
    ref = po.spatial_temporal_transformer(sd, "stt", x, ctx_t, _cfg(T))
    got = stt.to(DEV)(x.to(DEV), ctx_t.to(DEV)).cpu()
    err = (got - ref).abs()
    print(f"STT C={C} T={T} {h}x{w}: max {err.max().item():.3e} mean {err.mean().item():.3e} |ref| {ref.abs().max().item():.2f}")
    assert stt.precision == "precise"
    measured("stt_vs_oracle", C=C, T=T, max_abs=err.max().item(), mean_abs=err.mean().item(), ref_max=ref.abs().max().item())
    assert err.max().item() <= STT_TOL[0] and err.mean().item() <= STT_TOL[1]
    stt.precision = "fast"                     # emulation of the same graph: 1.75e-3 / 3.0e-4 at |out| ~ 7
    errf = (stt(x.to(DEV), ctx_t.to(DEV)).cpu() - ref).abs()
    measured("stt_vs_oracle_fast", C=C, max_abs=errf.max().item(), mean_abs=errf.mean().item())
    assert errf.max().item() <= STT_TOL_FAST[0] and errf.mean().item() <= STT_TOL_FAST[1]
