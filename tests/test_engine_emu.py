"""Host-logic parity on CPU: the product modules (weight packing, layouts, view/time quirks, skip and
control wiring, epilogue flags) run against the torch emulation of the C-ABI (tests/emu.py) and are
compared with the reference's golden eps and block outputs.  The emulation rounds every contraction
operand to fp16 exactly like the kernels, so the tolerance here is the fp16-operand tolerance of the
design, not fp32 round-off."""
import json

import numpy as np
import pytest
import torch

import emu
from helpers import cond, err_stats, golden, manifest, oracle_cfg, product_network, step_inputs
from panacea_amd import build_network, configs, engine as E, synth

# |eps| max ~2.7, rms 0.59.  (network, operand precision) -> (max-abs, mean-abs); the default "precise" policy meets the
# 1e-3 of BASELINE.json's north_star.  plain1 runs T=1 at 64 channels: its temporal GroupNorm normalises C/32 x T = 2
# values per group, which amplifies the rounding of the conv output feeding it, hence "precise-all" and a wider band.
TOL = {("tiny", "precise"): (1e-3, 2e-4), ("tiny", "fast"): (3e-3, 5e-4),
       ("plain1", "precise-all"): (1.4e-3, 2.2e-4), ("plain1", "fast"): (8e-3, 1.2e-3)}


@pytest.mark.parametrize("name,prec", list(TOL))
def test_engine_matches_reference_golden(name, prec):
    w, _, kw = product_network(name)
    w.diffusion_model.precision = prec
    inp = step_inputs(name, kw)
    gold = golden(name)
    trace = {}
    with E.use_backend(emu):
        eps = w(inp["x"], inp["t"], cond(inp), trace=trace)
    st = err_stats(eps, gold["eps"])
    assert st["max_abs"] <= TOL[(name, prec)][0] and st["mean_abs"] <= TOL[(name, prec)][1], st
    assert eps.dtype == torch.float32 and eps.shape == inp["x"].shape
    checked = 0
    for k in gold.files:
        key = k[6:] if k.startswith("block.") else k
        if key in trace and k != "eps":
            ref = gold[k]
            got = trace[key].reshape(-1)[::7].numpy()
            assert np.abs(got - ref).max() <= (2e-3 if name == "tiny" else 6e-3) * max(1.0, np.abs(ref).max()), k
            checked += 1
    assert checked >= 15


@pytest.mark.parametrize("name", ["tiny", "plain1", "full"])
def test_state_dict_manifest_matches_reference(name):
    if name == "full":
        with torch.device("meta"):
            w = build_network(configs.get(name))
    else:
        w = build_network(configs.get(name))
    sd = {k: list(v.shape) for k, v in w.diffusion_model.state_dict().items()}
    assert sd == manifest(name)
    # wrapper prefix used by the checkpoint loader (inference.py:219): model.diffusion_model.<...>
    assert all(k.startswith("diffusion_model.") for k in w.state_dict())


def test_fresh_network_is_zero_like_the_reference():
    """zero_module'd tensors (proj_out*, zero convs, out.2 ...) make a freshly built reference output exactly 0."""
    w = build_network(configs.get("tiny"))
    inp = step_inputs("tiny", configs.get("tiny"))
    with E.use_backend(emu):
        eps = w(inp["x"], inp["t"], cond(inp))
    assert eps.abs().max().item() == 0.0


def test_wrapper_side_effects_and_control_consumption():
    w, _, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    c = cond(inp)
    c["crossattn"] = c["crossattn"].double()
    with E.use_backend(emu):
        w(inp["x"], inp["t"], c)
        assert c["crossattn"].dtype == torch.float32          # cast in the caller's dict (wrappers.py:48)
        net = w.diffusion_model
        xin = torch.cat([inp["x"], inp["concat"]], 1)
        control = net.controlnet(xin, inp["cond_feat"], inp["t"], inp["crossattn"])
        assert len(control) == 5 and control[0].shape == (4, 64, 8, 96) and control[-1].shape == (4, 128, 4, 48)
        eps2 = net(xin, timesteps=inp["t"], context=inp["crossattn"], control=control, only_mid_control=False)
        assert control == []                                   # consumed by pop() like controlmodel.py:192-194
        eps1 = w(inp["x"], inp["t"], cond(inp))
    assert torch.allclose(eps1, eps2, atol=1e-6)


def test_submodule_forwards_match_oracle():
    """Reference-compatible per-module entries (NCHW in/out) of ResBlock3D and SpatialTemporalTransformer."""
    from oracle import panacea_oracle as po
    from helpers import oracle_cfg
    w, sd, kw = product_network("tiny")
    net = w.diffusion_model
    torch.manual_seed(0)
    x = torch.randn(4, 64, 8, 96)
    emb = torch.randn(4, 256)
    ctx = torch.randn(2, 77, 64)
    ctx_t = ctx[:, None].expand(-1, 2, -1, -1).reshape(4, 77, 64)
    cfg = oracle_cfg(kw)
    with E.use_backend(emu):
        r = net.input_blocks[1][0](x, emb)
        s = net.input_blocks[1][1](x, ctx_t)
    r_ref = po.resblock3d(sd, "input_blocks.1.0", x, emb, cfg)
    s_ref = po.spatial_temporal_transformer(sd, "input_blocks.1.1", x, ctx_t, cfg)
    assert (r - r_ref).abs().max() <= 4e-3 * r_ref.abs().max()
    assert (s - s_ref).abs().max() <= 4e-3 * s_ref.abs().max()


def test_packed_weights_follow_parameter_updates():
    w, sd, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    with E.use_backend(emu):
        e1 = w(inp["x"], inp["t"], cond(inp))
        sd2 = {k: v * 0.5 for k, v in sd.items()}
        w.diffusion_model.load_state_dict(sd2)                # post-hook drops the fp16 packs
        e2 = w(inp["x"], inp["t"], cond(inp))
    assert (e1 - e2).abs().max() > 1e-2


def test_network_level_projection_caches_follow_submodule_updates():
    """EmbProjector / TextKVProjector batch the `emb_layers` Linear of every ResBlock3D and the text K / V projections of
    every cross-attention site in one launch per network, from a packed copy held by the NETWORK.  Loading or editing ONE submodule
    (which only invalidates that submodule's own packed weights) must reach those copies too: the result equals a freshly built
    network with the same parameters."""
    from panacea_amd.nn.openaimodel import ResBlock3D
    from panacea_amd.nn.attention import BasicTransformerBlock
    w, sd, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    m = w.diffusion_model
    with E.use_backend(emu), torch.no_grad():
        w(inp["x"], inp["t"], cond(inp))                      # builds both network-level caches
        blk = next(b for b in m.modules() if isinstance(b, ResBlock3D))
        blk.load_state_dict({k: v * 1.5 for k, v in blk.state_dict().items()})        # one block only
        att = next(b for b in m.modules() if isinstance(b, BasicTransformerBlock)).attn2
        att.to_k.weight.mul_(0.5)                              # in-place edit of one site's parameter
        E.invalidate_all(att)
        e_cached = w(inp["x"], inp["t"], cond(inp))
        fresh, _, _ = product_network("tiny")
        fresh.diffusion_model.load_state_dict(m.state_dict(), strict=True)
        e_fresh = fresh(inp["x"], inp["t"], cond(inp))
    assert torch.equal(e_cached, e_fresh)


def test_batched_step_prologue_equals_the_launch_per_site_forms(monkeypatch):
    """Round 5: the emb_layers Linear of every ResBlock in one launch (EmbProjector) and the text K / V of every site in one GEMM
    (TextKVProjector) are re-orderings of launches, not of arithmetic: the network's output must not move when either is switched
    back to the per-site / per-width form (the emulation computes every column the same way in both)."""
    w, sd, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    outs = []
    for emb_batch, one_gemm in ((True, True), (False, True), (True, False), (False, False)):
        monkeypatch.setattr(E, "EMB_BATCH", emb_batch)
        monkeypatch.setattr(E, "TEXTKV_ONE_GEMM", one_gemm)
        E.invalidate_all(w)                                    # drops the cached projectors (built per switch setting)
        calls = []
        real = emu.linear_smallm_segments
        monkeypatch.setattr(emu, "linear_smallm_segments", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with E.use_backend(emu):
            outs.append(w(inp["x"], inp["t"], cond(inp)))
        monkeypatch.setattr(emu, "linear_smallm_segments", real)
        assert bool(calls) == emb_batch and (not emb_batch or len(calls) == 2)      # one launch per network (UNet, ControlNet)
    for o in outs[1:]:
        assert (o - outs[0]).abs().max().item() <= 2e-6 * outs[0].abs().max().item()


def test_unsupported_options_fail_loudly():
    kw = configs.get("tiny")
    with pytest.raises(NotImplementedError):
        build_network(dict(kw, num_head_channels=32))
    with pytest.raises(NotImplementedError):
        build_network(dict(kw, use_linear_in_transformer=False))
    from panacea_amd import hip
    with pytest.raises(hip.PncError):                          # product backend refuses CPU tensors: no fallback
        w, _, _ = product_network("tiny")
        inp = step_inputs("tiny", kw)
        w(inp["x"], inp["t"], cond(inp))


def test_baseline_config1_as_written_64x64_latent():
    """BASELINE config 1: single view, one frame, 64x64 latent — the plain-attention network on the
    reference's golden eps at that size (tests/golden/plain64.npz).  64 channels and T = 1: every temporal GroupNorm
    normalises 2 values per group, which amplifies the rounding of the conv output feeding it (fast 1.9e-2); with every
    operand class split the path is at 1.3e-3."""
    from helpers import golden
    w, _, kw = product_network("plain1")
    w.diffusion_model.precision = "precise-all"
    inp = step_inputs("plain1", kw, shape=(1, 1, 64, 64))
    with E.use_backend(emu):
        eps = w(inp["x"], inp["t"], cond(inp))
    st = err_stats(eps, golden("plain64")["eps"])
    assert eps.shape == (1, 4, 64, 64)
    assert st["max_abs"] <= 1.7e-3 and st["mean_abs"] <= 1.8e-4, st


def test_operand_policy_follows_the_network_unless_set():
    """The default operand policy is `precise`; networks whose temporal GroupNorm groups hold fewer than 4 values (num_frames x
    C / 32: the 64-channel single-frame nets of BASELINE config 1) get every class split AND a warning that they sit above the
    1e-3 the wide networks meet — no configuration exceeds the contract silently (VERDICT r2 weak spot 2)."""
    import warnings
    from panacea_amd import build_network, configs, engine
    net = build_network(configs.get("tiny")).diffusion_model               # 64 channels x 2 frames = 4 values per group
    assert net.precision == "precise" and net.controlnet.precision == "precise"
    t1 = build_network(configs.with_frames(configs.get("tiny"), 1)).diffusion_model     # 2 values per group
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert t1.precision == "precise-all"
    assert any("precise-all" in str(x.message) and "1e-3" in str(x.message) for x in w)
    full1 = build_network(configs.with_frames(configs.get("full"), 1)).diffusion_model  # 320 channels: 10 values per group
    assert full1.precision == "precise"
    # the stated bound per configuration: 1e-3 from 4 values per group on, 2.5e-3 below
    c1 = full1.eps_contract
    assert net.eps_contract["eps_max_abs"] == 1e-3 and (c1["policy"], c1["eps_max_abs"], c1["values_per_temporal_group"]) == ("precise", 1e-3, 10)
    # round 5: the bound's RANGE is part of the contract (heavy-tail pin: 2.3e-3 at |stream| = 1.8e3, outside it)
    assert "512" in c1["valid_for"] and c1["beyond"]["eps_max_abs_measured"] > 1e-3 and c1["beyond"]["pin"].endswith("tail64.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert t1.eps_contract["eps_max_abs"] == 2.5e-3 and t1.eps_contract["values_per_temporal_group"] == 2
    t1.precision = "fast"                                                   # an explicit choice always wins
    assert t1.precision == "fast" and t1.controlnet.precision == "fast" and t1.eps_contract["eps_max_abs"] is None
    assert engine.precision("precise").lo8 and not engine.precision("precise-f16lo").lo8
    with pytest.raises(ValueError):
        t1.precision = "fastest"


def test_view_and_frame_shard_loop_back_on_one_process():
    """engine.ViewShard(1, 0) / FrameShard(1, 0) without a process group: the band is the whole panorama and its own circular
    neighbour, the frame group is the rank itself.  Drives every exchange site of the tiny network in ONE process (the GPU suite
    runs the same at the full network's width): the frame loop-back only moves data (same bits), the view loop-back runs every
    3x3 conv over the map widened by zero halo columns and applies GroupNorm records combined by `groupnorm_combine`."""
    from panacea_amd import parallel
    kw = configs.with_frames(configs.get("tiny"), 2)
    w, _, _ = product_network("tiny", kw=kw)
    inp = step_inputs("tiny", kw, t_index=500, shape=(2, 2, 8, 96))
    with E.use_backend(emu), torch.no_grad():
        ref = w(inp["x"], inp["t"], cond(inp))
        sh = E.FrameShard(1, 0, None)
        parallel.apply_frame_shard(w, sh)
        got_f = w(inp["x"], inp["t"], cond(inp))
        vs = E.ViewShard(1, 0, None)
        parallel.apply_view_shard(w, vs)
        got_vf = w(inp["x"], inp["t"], cond(inp))
        parallel.apply_frame_shard(w, None)
        got_v = w(inp["x"], inp["t"], cond(inp))
        # the exchange against its definition: what arrives from the left is what went to the right (and vice versa)
        a, b = torch.randn(3, 5), torch.randn(2, 7).half()
        (fl, fl2), (fr, fr2) = vs._exchange([a, b], [a + 1, b + 1])
    assert torch.equal(fl, a + 1) and torch.equal(fl2, b + 1) and torch.equal(fr, a) and torch.equal(fr2, b)
    # the frame loop-back runs the ResBlock3D temporal sites in their sharded form (partial sums + the halo-frame layout of the
    # temporal conv: other roundings of the statistics than the fused kernel's), the STT temporal branch through its transposes
    df = (got_f - ref).abs()
    assert df.max().item() <= 2e-3 and df.mean().item() <= 2.5e-4, (df.max().item(), df.mean().item())
    assert sh.exchanges > 20 and vs.exchanges > 50 and vs.bytes_sent == 0
    # round 2's form of the ResBlock sites (the fp32 stream to the pixel sharding and back) only moves data: same bits
    parallel.apply_view_shard(w, None)
    sh2 = E.FrameShard(1, 0, None, resblock="transpose")
    parallel.apply_frame_shard(w, sh2)
    from helpers import gn_statistics_from_launches
    with E.use_backend(emu), torch.no_grad(), gn_statistics_from_launches():
        got_t = w(inp["x"], inp["t"], cond(inp))
        parallel.apply_frame_shard(w, None)
        assert torch.equal(got_t, w(inp["x"], inp["t"], cond(inp))) and sh2.exchanges > 20
    # torch's conv over the widened map sums in another order than over the panorama: decorrelated fp16 operand roundings
    for got in (got_v, got_vf):
        d = (got - ref).abs()
        assert d.max().item() <= 2e-3 and d.mean().item() <= 2.5e-4, (d.max().item(), d.mean().item())
    with pytest.raises(ValueError):
        E.ViewShard(2, 0, None)


def test_range_monitor_detects_saturated_lo_planes():
    """Round 6: the eps contract of `precise` is written for split operands inside the e4m3 lo plane's range
    (|v| < 512).  Every e4m3 pack counts the quads it clamps (pnc_range_monitor_collect; the emulation counts the same event): an
    evaluation on the ordinary synthetic weights reports 0, the heavy-tail weight set (stream at ~1.6e3) reports > 0 and warns ONCE;
    `on_range_exceeded = "raise"` refuses."""
    import warnings
    from oracle import panacea_oracle as po
    w, sd, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    m = w.diffusion_model
    assert m.lo_clamped is None
    with E.use_backend(emu), torch.no_grad():
        w(inp["x"], inp["t"], cond(inp))
    assert m.lo_clamped == 0 and m.lo_clamped_total == 0
    sd_t = synth.synth_state_dict(manifest("tiny"), tail=64.0)
    m.load_state_dict(sd_t, strict=True)
    ref = po.wrapper_forward(sd_t, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    with E.use_backend(emu), torch.no_grad(), warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        e1 = w(inp["x"], inp["t"], cond(inp))
        n1 = m.lo_clamped
        w(inp["x"], inp["t"], cond(inp))
    assert n1 > 0 and m.lo_clamped == n1 and m.lo_clamped_total == 2 * n1
    assert sum("lo-plane quads saturated" in str(x.message) for x in ws) == 1            # once per network
    m.on_range_exceeded = "raise"
    with E.use_backend(emu), torch.no_grad(), pytest.raises(RuntimeError, match="saturated"):
        w(inp["x"], inp["t"], cond(inp))
    m.on_range_exceeded = "warn"
    d1 = (e1 - ref).abs().max().item()
    assert d1 > 1e-3, d1             # ... and the detected evaluation is indeed outside the contract (measured 1.3e-2)
