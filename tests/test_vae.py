"""First-stage decoder: oracle pinned against the reference's own Decoder (tests/golden/vae_tiny.npz,
oracle/gen_golden_vae.py), host logic of the mirror module against the emulated C-ABI on CPU, and the HIP path on the
GPU (kernel parity of the row softmax, decoder parity, full-size decode)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import emu
from oracle import vae_oracle as vo
from panacea_amd import engine as E, synth
from panacea_amd.nn import model

GOLDEN = Path(__file__).resolve().parent / "golden"
TINY = dict(double_z=True, z_channels=4, resolution=16, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2], num_res_blocks=1,
            attn_resolutions=[], dropout=0.0)
FULL = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
            num_res_blocks=2, attn_resolutions=[], dropout=0.0)          # configs/inference_nuscenes.yaml:101-111
DEV = "cuda"


def _tiny():
    man = json.loads((GOLDEN / "manifest_vae_tiny.json").read_text())
    sd = synth.synth_state_dict(man)
    g = np.load(GOLDEN / "vae_tiny.npz")
    return man, sd, g


def test_oracle_matches_reference_decoder():
    _, sd, g = _tiny()
    tr = {}
    img = vo.decode(sd, vo.VaeConfig(ch=64, ch_mult=[1, 2], num_res_blocks=1), torch.from_numpy(g["z"]), trace=tr)
    assert np.abs(img.numpy() - g["img"]).max() < 2e-5
    assert np.abs(tr["mid"].reshape(-1)[::7].numpy() - g["mid_s7"]).max() < 2e-5
    assert float(g["oracle_vs_reference"]) < 1e-5 and np.abs(g["img"]).max() > 1.0


def test_mirror_module_tree_and_host_logic_vs_reference():
    man, sd, g = _tiny()
    fs = model.FirstStageDecoder(4, TINY)
    assert {k: list(v.shape) for k, v in fs.state_dict().items()} == man          # names + shapes of the reference
    fs.load_state_dict(sd, strict=True)
    with E.use_backend(emu):
        img = fs.decode(torch.from_numpy(g["z"]))
    d = np.abs(img.numpy() - g["img"])
    assert img.shape == (2, 3, 16, 96) and d.max() < 8e-3 and d.mean() < 1e-3, (d.max(), d.mean())
    # unsupported options fail loudly instead of computing something else
    with pytest.raises(NotImplementedError):
        model.Decoder(**dict(TINY, use_linear_attn=True))
    with pytest.raises(NotImplementedError):
        model.ResnetBlock(in_channels=64, out_channels=64, temb_channels=512)
    # a full-engine checkpoint sub-tree loads with strict=False (encoder / quant_conv keys are simply unexpected)
    extra = dict(sd)
    extra["encoder.conv_in.weight"] = torch.zeros(64, 3, 3, 3)
    extra["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    res = fs.load_state_dict(extra, strict=False)
    assert res.missing_keys == [] and sorted(res.unexpected_keys) == ["encoder.conv_in.weight", "quant_conv.weight"]


def _tiny_enc():
    man = json.loads((GOLDEN / "manifest_vae_enc_tiny.json").read_text())
    return man, synth.synth_state_dict(man), np.load(GOLDEN / "vae_enc_tiny.npz")


def test_encoder_oracle_and_host_logic_vs_reference():
    man, sd, g = _tiny_enc()
    x = torch.from_numpy(g["x"])
    tr = {}
    mom = vo.encode_moments(sd, vo.VaeConfig(ch=64, ch_mult=[1, 2], num_res_blocks=1), x, trace=tr)
    assert np.abs(mom.numpy() - g["moments"]).max() < 2e-5
    assert np.abs(tr["down.0"].reshape(-1)[::7].numpy() - g["down0_s7"]).max() < 2e-5
    fe = model.FirstStageEncoder(4, TINY)
    assert {k: list(v.shape) for k, v in fe.state_dict().items()} == man
    fe.load_state_dict(sd, strict=True)
    with E.use_backend(emu):
        got = fe.moments(x)
        z_mean = fe.encode(x, sample=False)
        z1 = fe.encode(x, generator=torch.Generator().manual_seed(1))
    d = np.abs(got.numpy() - g["moments"])
    assert got.shape == (2, 8, 8, 48) and d.max() < 8e-3 and d.mean() < 1e-3, (d.max(), d.mean())
    assert torch.equal(z_mean, got[:, :4]) and z1.shape == (2, 4, 8, 48) and not torch.equal(z1, z_mean)
    with pytest.raises(NotImplementedError):
        model.Downsample(64, with_conv=False)


@pytest.mark.gpu
def test_encoder_hip_matches_reference_golden_and_roundtrip_shapes():
    man, sd, g = _tiny_enc()
    fe = model.FirstStageEncoder(4, TINY)
    fe.load_state_dict(sd, strict=True)
    fe = fe.to(DEV)
    got = fe.moments(torch.from_numpy(g["x"]).to(DEV))
    torch.cuda.synchronize()
    d = (got.cpu() - torch.from_numpy(g["moments"])).abs()
    print("vae encoder tiny vs reference:", d.max().item(), d.mean().item())
    assert d.max().item() < 8e-3 and d.mean().item() < 1e-3
    assert torch.equal(got, fe.moments(torch.from_numpy(g["x"]).to(DEV)))
    # the nuScenes first stage on one 256x768 strip (mid attention over 32x96 tokens), against the oracle
    fe = model.FirstStageEncoder(4, FULL)
    fman = {k: list(v.shape) for k, v in fe.state_dict().items()}
    fsd = synth.synth_state_dict(fman)
    fe.load_state_dict(fsd, strict=True)
    x = torch.tanh(torch.randn(1, 3, 128, 384, generator=torch.Generator().manual_seed(8)))
    ref = vo.encode_moments(fsd, vo.VaeConfig(), x)
    got = fe.to(DEV).moments(x.to(DEV))
    d = (got.cpu() - ref).abs()
    print("vae encoder full-width, 128x384 image vs oracle:", d.max().item(), d.mean().item(), "ref max", ref.abs().max().item())
    assert got.shape == (1, 8, 16, 48) and d.max().item() < 2e-2 and d.mean().item() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("F,Hin,Win,Cin,N", [(2, 8, 24, 64, 64), (1, 16, 48, 128, 128), (1, 6, 10, 8, 32)])
def test_conv3x3_stride2_bottom_right_padding(F, Hin, Win, Cin, N):
    from panacea_amd import hip
    Hout, Wout = Hin // 2, Win // 2
    M, K = F * Hout * Wout, 9 * Cin
    g = torch.Generator().manual_seed(Cin + Hin)
    x = torch.randn(F, Hin, Win, Cin, generator=g).half().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    conv = dict(Cin=Cin, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, stride=2, upsample=0, pad_br=1)
    oh, oe = torch.zeros(M, N, device=DEV), torch.zeros(M, N, device=DEV)
    hip.gemm(x, w, M=M, N=N, K=K, a_mode=hip.A_CONV3X3, conv=conv, bias=bias, out32=oh, ldc32=N)
    emu.gemm(x, w, M=M, N=N, K=K, a_mode=emu.A_CONV3X3, conv=conv, bias=bias, out32=oe, ldc32=N)
    torch.cuda.synchronize()
    assert (oh - oe).abs().max().item() < 3e-3
    # it is NOT the symmetric-padding stride-2 conv
    conv_sym = dict(conv, pad_br=0)
    emu.gemm(x, w, M=M, N=N, K=K, a_mode=emu.A_CONV3X3, conv=conv_sym, bias=bias, out32=oe, ldc32=N)
    assert (oh - oe).abs().max().item() > 1e-1


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,lds,scale", [(37, 384, 384, 0.125), (5, 12288, 12288, 512 ** -0.5), (64, 1000, 1024, 1.0),
                                           (3, 16384, 16384, 0.05), (9, 4, 8, 2.0)])
def test_softmax_rows_kernel(M, N, lds, scale):
    from panacea_amd import hip
    g = torch.Generator().manual_seed(N)
    s = (torch.randn(M, lds, generator=g) * 30).to(DEV)
    ph = torch.zeros(M, lds, device=DEV, dtype=torch.float16)
    pe = torch.zeros_like(ph)
    hip.softmax_rows(s, lds, M, N, scale, ph, lds)
    emu.softmax_rows(s, lds, M, N, scale, pe, lds)
    torch.cuda.synchronize()
    assert torch.isfinite(ph).all() and (ph[:, N:] == 0).all()
    assert (ph[:, :N].float() - pe[:, :N].float()).abs().max().item() <= 1e-3
    assert (ph[:, :N].float().sum(-1) - 1).abs().max().item() < 2e-2
    ph2 = torch.zeros_like(ph)
    hip.softmax_rows(s, lds, M, N, scale, ph2, lds)
    torch.cuda.synchronize()
    assert torch.equal(ph, ph2)


@pytest.mark.gpu
def test_decoder_hip_matches_reference_golden():
    _, sd, g = _tiny()
    fs = model.FirstStageDecoder(4, TINY)
    fs.load_state_dict(sd, strict=True)
    fs = fs.to(DEV)
    img = fs.decode(torch.from_numpy(g["z"]).to(DEV))
    torch.cuda.synchronize()
    d = (img.cpu() - torch.from_numpy(g["img"])).abs()
    print("vae tiny vs reference:", d.max().item(), d.mean().item())
    assert d.max().item() < 8e-3 and d.mean().item() < 1e-3
    assert torch.equal(img, fs.decode(torch.from_numpy(g["z"]).to(DEV)))            # run-to-run bit-identical


@pytest.mark.gpu
def test_decoder_full_size_against_oracle_on_a_crop_and_timing():
    """The nuScenes first stage (ch 128, mult 1-2-4-4, 2 res blocks, mid attention): 2 frames of a 16x48 latent against
    the oracle, then 8 frames at the real 32x384 latent -> 256x3072 panorama (12288-token attention per frame)."""
    import time
    fs = model.FirstStageDecoder(4, FULL)
    man = {k: list(v.shape) for k, v in fs.state_dict().items()}
    sd = synth.synth_state_dict(man)
    fs.load_state_dict(sd, strict=True)
    z = torch.randn(2, 4, 16, 48, generator=torch.Generator().manual_seed(3)) * 2.0
    ref = vo.decode(sd, vo.VaeConfig(), z)
    fs = fs.to(DEV)
    img = fs.decode(z.to(DEV))
    d = (img.cpu() - ref).abs()
    print("vae full-width, 16x48 latent vs oracle:", d.max().item(), d.mean().item(), "ref max", ref.abs().max().item())
    assert d.max().item() < 2e-2 and d.mean().item() < 2e-3
    z8 = (torch.randn(8, 4, 32, 384, generator=torch.Generator().manual_seed(4)) * 2.0).to(DEV)
    out = fs.decode(z8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fs.decode(z8)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out.shape == (8, 3, 256, 3072) and torch.isfinite(out).all()
    print(f"decode of 8 frames at 256x3072: {dt * 1e3:.1f} ms  (~62 TFLOP => {62.0 / dt:.0f} TFLOP/s)")


def _tiny_pipeline(device):
    from helpers import product_network, step_inputs
    w, _, kw = product_network("tiny", device)
    inp = step_inputs("tiny", kw, device)
    T = kw["num_frames"]
    c = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
    uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
    _, sd, _ = _tiny()
    fs = model.FirstStageDecoder(4, TINY)
    fs.load_state_dict(sd, strict=True)
    return w, fs.to(device), c, uc, inp["x"][T:]


def test_pipeline_latents_to_frames_composes_on_the_emulated_backend(tmp_path):
    from panacea_amd import checkpoint as ck, pipeline
    w, fs, c, uc, noise = _tiny_pipeline("cpu")
    with E.use_backend(emu):
        a = pipeline.sample_frames(w, fs, c, uc, noise, num_steps=2, hoist=True)
        b = pipeline.sample_frames(w, fs, c, uc, noise, num_steps=2, hoist=False)
    assert a.shape == (noise.shape[0], 3, 8 * 2, 96 * 2) and torch.isfinite(a).all() and torch.equal(a, b)
    assert len(ck.save_view_frames(a, str(tmp_path), [f"v{i}" for i in range(6)], view_width=a.shape[-1] // 6)) == 6 * a.shape[0]


@pytest.mark.gpu
def test_pipeline_latents_to_frames_on_the_gpu():
    from panacea_amd import pipeline
    w, fs, c, uc, noise = _tiny_pipeline(DEV)
    a = pipeline.sample_frames(w, fs, c, uc, noise, num_steps=3)
    b = pipeline.sample_frames(w, fs, c, uc, noise, num_steps=3, hoist=False)
    torch.cuda.synchronize()
    assert a.shape == (noise.shape[0], 3, 16, 192) and torch.isfinite(a).all() and torch.equal(a, b)


def test_full_width_first_stage_host_logic_on_the_emulated_backend():
    """The nuScenes first-stage architecture (4 levels, 2 res blocks, 128..512 channels, mid attention) on a small image:
    decoder and encoder mirror modules vs the oracle through the emulated C-ABI (the GPU versions of this test run the
    same modules through the HIP library)."""
    torch.manual_seed(0)
    fd = model.FirstStageDecoder(4, FULL)
    sd = synth.synth_state_dict({k: list(v.shape) for k, v in fd.state_dict().items()})
    fd.load_state_dict(sd, strict=True)
    z = torch.randn(1, 4, 8, 24) * 2.0
    ref = vo.decode(sd, vo.VaeConfig(), z)
    with E.use_backend(emu):
        img = fd.decode(z)
    d = (img - ref).abs()
    assert img.shape == (1, 3, 64, 192) and d.max().item() < 2e-2 and d.mean().item() < 2e-3, (d.max().item(), d.mean().item())
    fe = model.FirstStageEncoder(4, FULL)
    esd = synth.synth_state_dict({k: list(v.shape) for k, v in fe.state_dict().items()})
    fe.load_state_dict(esd, strict=True)
    x = torch.tanh(torch.randn(1, 3, 64, 192))
    eref = vo.encode_moments(esd, vo.VaeConfig(), x)
    with E.use_backend(emu):
        mom = fe.moments(x)
    d = (mom - eref).abs()
    assert mom.shape == (1, 8, 8, 24) and d.max().item() < 2e-2 and d.mean().item() < 2e-3, (d.max().item(), d.mean().item())
