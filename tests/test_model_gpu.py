"""End-to-end parity of the HIP path on the MI355X, through the drop-in boundary
(`OpenAIWrapperControlLDM3D.forward` -> C-ABI kernels):

  * tiny / plain1 networks vs the reference's golden eps (tests/golden) and block outputs;
  * the FULL Panacea+ network (2 478 tensors, C = 320..1280, 5..20 heads) on a small panorama vs the CPU
    oracle on the same synthetic weights and inputs;
  * size-independent properties at BASELINE config 3 size (16 panoramic frames of 32x384): the two CFG
    halves are independent, a frame permutation inside a sample permutes the output (temporal
    attention/conv are the only frame couplings and the text context is per sample), determinism.

Tolerances, eps max-abs against the fp32 reference / oracle on identical fp16-representable weights:
  * precision "precise" (the default of the product: split fp16 operands where the error budget needs them):
    **1e-3**, the tolerance BASELINE.json's north_star states, on every 8-/4-/2-frame configuration;
  * precision "fast" (plain fp16 operands): pinned ~1.3x above the measured values so that a regression trips.
"""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, cond, err_stats, golden, manifest, measured, oracle_cfg, product_network, step_inputs
from oracle import panacea_oracle as po
from panacea_amd import configs, hip

pytestmark = pytest.mark.gpu
DEV = "cuda"
NORTH_STAR = 1e-3        # eps max-abs error stated by BASELINE.json north_star
CONFIG2_TOL = (NORTH_STAR, 2e-4)
# (max-abs, mean-abs) per (network, precision).  measured r2: tiny 7.7e-4 / 1.3e-4 precise, 1.9e-3 / 3.5e-4 fast;
# plain1 (T = 1, 64 channels: the temporal GroupNorm normalises 2 values per group, which amplifies the rounding of the
# conv3x3 output feeding it) 1.0e-3 / 1.6e-4 with every class split, 5.7e-3 / 8.4e-4 fast
TOL = {("tiny", "precise"): (NORTH_STAR, 2e-4), ("tiny", "fast"): (3e-3, 5e-4),
       ("plain1", "precise-all"): (1.4e-3, 2.2e-4), ("plain1", "fast"): (8e-3, 1.2e-3)}


# block outputs / ControlNet residuals (stride-7 samples of the reference's tensors): max-abs error relative to max(1, max|ref|)
# of the tensor, pinned ~1.5x above the worst block measured on the MI355X in round 3 (gpurun_out/test_measurements.log:
# 5.9e-4 / 7.2e-4 / 8.8e-4 / 4.0e-3; round 2 allowed 2e-3 on tiny and 6e-3 on plain1 whatever the policy)
BLOCK_TOL = {("tiny", "precise"): 9e-4, ("tiny", "fast"): 1.1e-3, ("plain1", "precise-all"): 1.4e-3, ("plain1", "fast"): 6e-3}


def test_library_is_loaded_and_native():
    lib = hip.load()
    assert b"gfx950" in lib.pnc_version()
    maps = open("/proc/self/maps").read()
    assert "libpanacea_hip.so" in maps


@pytest.mark.parametrize("name,prec", list(TOL))
def test_hip_path_matches_reference_golden(name, prec):
    w, _, kw = product_network(name, DEV)
    w.diffusion_model.precision = prec
    inp = step_inputs(name, kw, DEV)
    gold = golden(name)
    trace = {}
    eps = w(inp["x"], inp["t"], cond(inp), trace=trace)
    torch.cuda.synchronize()
    st = err_stats(eps, gold["eps"])
    print(name, prec, st)
    assert eps.is_cuda and eps.dtype == torch.float32
    assert st["max_abs"] <= TOL[(name, prec)][0] and st["mean_abs"] <= TOL[(name, prec)][1], st
    worst = 0.0
    for k in gold.files:
        key = k[6:] if k.startswith("block.") else k
        if key in trace and k != "eps":
            ref = gold[k]
            got = trace[key].reshape(-1)[::7].cpu().numpy()
            rel = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, rel)
            assert rel <= BLOCK_TOL[(name, prec)], (k, rel)
    measured("block_trace", net=name, prec=prec, worst_rel=float(worst))


def test_baseline_config1_64x64_latent_vs_reference_golden():
    """BASELINE config 1 as written (single view, 1 frame, 64x64 latent) on the GPU against the reference's golden eps;
    tolerance and its reason: tests/test_engine_emu.py::test_baseline_config1_as_written_64x64_latent."""
    w, _, kw = product_network("plain1", DEV)
    w.diffusion_model.precision = "precise-all"
    inp = step_inputs("plain1", kw, DEV, shape=(1, 1, 64, 64))
    eps = w(inp["x"], inp["t"], cond(inp))
    st = err_stats(eps, golden("plain64")["eps"])
    print("plain 64x64:", st)
    assert st["max_abs"] <= 2.2e-3 and st["mean_abs"] <= 2.2e-4, st
    # the bound the product states for this configuration (2 values per temporal GroupNorm group: INTEGRATION.md section 1)
    assert st["max_abs"] <= w.diffusion_model.eps_contract["eps_max_abs"] == 2.5e-3


def test_hip_path_other_timesteps_and_frames_vs_oracle():
    """tiny network, T = 4 frames, t = 333: outside the golden vectors, against the oracle directly."""
    kw = configs.with_frames(configs.get("tiny"), 4)
    w, sd, _ = product_network("tiny", DEV, kw=kw)
    inp = step_inputs("tiny", kw, "cpu", t_index=333, shape=(2, 4, 8, 96))
    ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    g = {k: v.to(DEV) for k, v in inp.items()}
    eps = w(g["x"], g["t"], cond(g))
    st = err_stats(eps, ref)
    print(st)
    assert w.diffusion_model.precision == "precise"
    assert st["max_abs"] <= NORTH_STAR and st["mean_abs"] <= 2e-4, st


def test_hip_path_single_frame_six_views_vs_oracle():
    """BASELINE config 2 shape class on the 64-channel tiny net: num_frames = 1 (temporal attention over one frame, conv1d
    with both neighbours padded) and a CFG batch of 2.  At 64 channels the T = 1 temporal GroupNorm normalises 2 values per
    group, which amplifies the rounding of the conv3x3 output feeding it (same band as `plain1`): every class split gives
    1.1e-3, the default precise set 4.0e-3; the full-width T = 1 test below (10-40 values per group) is the config-2 gate."""
    kw = configs.with_frames(configs.get("tiny"), 1)
    w, sd, _ = product_network("tiny", DEV, kw=kw)
    w.diffusion_model.precision = "precise-all"
    inp = step_inputs("tiny", kw, "cpu", t_index=666, shape=(2, 1, 8, 96))
    ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    g = {k: v.to(DEV) for k, v in inp.items()}
    eps = w(g["x"], g["t"], cond(g))
    st = err_stats(eps, ref)
    print("tiny, T=1, 6 views:", st)
    assert st["ref_max"] > 1.0
    assert st["max_abs"] <= 1.5e-3 and st["mean_abs"] <= 2e-4, st


def test_full_width_single_frame_config2_vs_oracle():
    """BASELINE config 2 at FULL width: the Panacea+ network with num_frames = 1 (6 views, intra-view + cross-view
    attention; the temporal branch degenerates to one frame), CFG batch 2, latent 16x192, against the oracle."""
    kw = configs.with_frames(configs.get("full"), 1)
    w, sd, _ = product_network("full", "cpu", kw=kw)
    inp = step_inputs("full", kw, "cpu", shape=(2, 1, 16, 192))
    ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    w = w.to(DEV)
    g = {k: v.to(DEV) for k, v in inp.items()}
    eps = w(g["x"], g["t"], cond(g))
    st = err_stats(eps, ref)
    print("full network, T=1, 16x192:", w.diffusion_model.precision, st)
    assert st["ref_max"] > 1.0
    assert st["max_abs"] <= CONFIG2_TOL[0] and st["mean_abs"] <= CONFIG2_TOL[1], st
    # ... and at the workload BASELINE config 2 names — 6 views x 1 frame at 256x512 = a 32x384 latent, CFG batch 2 — against
    # the REFERENCE's own forward (round 4 pin: oracle/gen_golden_full.py --frames 1 -> tests/golden/full_cfg2.npz)
    g2 = np.load(GOLDEN / "full_cfg2.npz")
    gi = step_inputs("full", kw, DEV, shape=(2, 1, 32, 384))
    st2 = pin_stats(w(gi["x"], gi["t"], cond(gi)), g2)               # (the whole eps since round 6)
    print("config 2 (T=1, 32x384) vs reference:", st2)
    measured("full_cfg2", max_abs=st2["max_abs"], mean_abs=st2["mean_abs"], elements=st2["elements"])
    assert st2["max_abs"] <= CONFIG2_TOL[0] and st2["mean_abs"] <= CONFIG2_TOL[1], st2


def test_full_network_small_panorama_vs_oracle():
    """Every tensor of the Panacea+ stage-2 network at its real width; latent 16x192 (L2 views 4x8, L3 views
    2x4 tokens: exercises the narrow-view gather of the attention kernel), B=1, T=2."""
    kw = configs.with_frames(configs.get("full"), 2)
    w, sd, _ = product_network("full", "cpu", kw=kw)
    inp = step_inputs("full", kw, "cpu", shape=(1, 2, 16, 192))
    ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    w = w.to(DEV)
    g = {k: v.to(DEV) for k, v in inp.items()}
    eps = w(g["x"], g["t"], cond(g))
    st = err_stats(eps, ref)
    print("full network, 16x192:", st)
    assert st["ref_max"] > 1.0
    assert st["max_abs"] <= NORTH_STAR and st["mean_abs"] <= 2e-4, st          # measured 6.4e-4 / 1.2e-4
    w.diffusion_model.precision = "fast"
    st = err_stats(w(g["x"], g["t"], cond(g)), ref)
    print("full network, 16x192, fast:", st)
    assert st["max_abs"] <= 3.4e-3 and st["mean_abs"] <= 5e-4, st               # measured 2.6e-3 / 3.6e-4


def pin_stats(eps: torch.Tensor, gold) -> dict:
    """eps of the HIP path against a full-size pin of the reference's forward: over ALL elements where the pin holds the whole eps
    (key "eps": full_cfg3 / full_cfg3_t500 / full_cfg5_step0 since round 6, SURVEY 8c "full-size pin"), else over its stride-7 sample"""
    if "eps" in gold.files:
        st = err_stats(eps, gold["eps"])
        st["elements"] = "all"
        assert np.array_equal(gold["eps"].reshape(-1)[::7], gold["eps_s7"])
    else:
        st = err_stats(eps.reshape(-1)[::7], gold["eps_s7"])
        st["elements"] = "stride-7 sample"
    return st


@pytest.fixture(scope="module")
def full_net():
    w, _, kw = product_network("full", "cpu")
    return w.to(DEV), kw


def test_full_size_properties_and_golden(full_net):
    """BASELINE config 3 size: (B, T) = (2, 8), 32x384 latent, 256x3072 BEV hint."""
    w, kw = full_net
    inp = step_inputs("full", kw, DEV)
    eps = w(inp["x"], inp["t"], cond(inp))
    torch.cuda.synchronize()
    assert eps.shape == (16, 4, 32, 384) and torch.isfinite(eps).all()
    # run-to-run: bit-identical (no float atomics anywhere on the path)
    eps_b = w(inp["x"], inp["t"], cond(inp))
    assert torch.equal(eps, eps_b)
    # the CFG halves never interact inside the network (guiders.py:31-40): the cond half alone gives the same eps
    half = {k: (v[8:] if v.shape[0] == 16 else v[1:]).contiguous() for k, v in inp.items()}
    eps_h = w(half["x"], half["t"], cond(half))
    assert torch.equal(eps_h, eps[8:])
    # samples are isolated: perturbing sample 0 (latent, hint, text) leaves sample 1's eps unchanged
    pert = {k: v.clone() for k, v in inp.items()}
    pert["x"][:8] += 0.5
    pert["cond_feat"][:8] = 1.0 - pert["cond_feat"][:8]
    pert["crossattn"][0] *= -1.0
    eps_p = w(pert["x"], pert["t"], cond(pert))
    assert torch.equal(eps_p[8:], eps[8:])
    assert (eps_p[:8] - eps[:8]).abs().max().item() > 1e-2
    # full-size golden: the REFERENCE itself on the same synthetic weights/inputs (oracle/gen_golden_full.py)
    path = GOLDEN / "full_cfg3.npz"
    if path.exists():
        g = np.load(path)
        st = pin_stats(eps, g)
        print("config 3 vs reference:", w.diffusion_model.precision, st)
        measured("full_cfg3", t=999, salt=0, max_abs=st["max_abs"], mean_abs=st["mean_abs"], elements=st["elements"])
        assert w.diffusion_model.precision == "precise"
        assert st["max_abs"] <= NORTH_STAR and st["mean_abs"] <= 2e-4, st
        assert st["elements"] == "all"                     # round 6: this pin holds the whole eps (3.1 MB fp32)
        assert w.diffusion_model.lo_clamped == 0           # range monitor: the evaluation stayed inside the contract's range
        # further pins of the reference's own forward (round 3): other noise levels, another input seed — same gate
        for fname, t_index, salt in (("full_cfg3_t500.npz", 500, 0), ("full_cfg3_t39_s1.npz", 39, 1)):
            from panacea_amd import synth
            gi = {k: v.to(DEV) for k, v in synth.synth_inputs(2, 8, 32, 384, context_dim=kw["context_dim"], t_index=t_index,
                                                               salt=salt).items()}
            gp = np.load(GOLDEN / fname)
            assert int(gp["t_index"]) == t_index and int(gp["input_salt"]) == salt
            st2 = pin_stats(w(gi["x"], gi["t"], cond(gi)), gp)
            print(f"config 3 vs reference, t={t_index} salt={salt}:", st2)
            measured("full_cfg3", t=t_index, salt=salt, max_abs=st2["max_abs"], mean_abs=st2["mean_abs"], elements=st2["elements"])
            assert t_index != 500 or st2["elements"] == "all"
            assert st2["max_abs"] <= NORTH_STAR and st2["mean_abs"] <= 2e-4, (fname, st2)
            del gi
        # BASELINE config 5 (round 4 pin): the reference's own forward on the YAML-exact inputs of sampler step 0 — last-frame
        # `concat` conditioning (final_cond_zero), share-noise latent, t = 999 (oracle/gen_golden_full.py --yaml-exact)
        from panacea_amd import synth
        g5 = np.load(GOLDEN / "full_cfg5_step0.npz")
        gi = {k: v.to(DEV) for k, v in synth.yaml_exact_step0_inputs(8, 32, 384, context_dim=kw["context_dim"]).items()}
        st5 = pin_stats(w(gi["x"], gi["t"], cond(gi)), g5)
        print("config 5 (yaml-exact step 0) vs reference:", st5)
        measured("full_cfg5_step0", max_abs=st5["max_abs"], mean_abs=st5["mean_abs"], elements=st5["elements"])
        assert st5["max_abs"] <= NORTH_STAR and st5["mean_abs"] <= 2e-4, st5
        del gi
        w.diffusion_model.precision = "fast"
        st = err_stats(w(inp["x"], inp["t"], cond(inp)).reshape(-1)[::7], g["eps_s7"])
        w.diffusion_model.precision = "precise"
        print("config 3 vs reference, fast:", st)
        assert st["max_abs"] <= 3.0e-3 and st["mean_abs"] <= 4.7e-4, st           # measured 2.3e-3 / 3.6e-4
    else:
        pytest.skip("tests/golden/full_cfg3.npz not generated yet (property checks passed)")


# Round 5 (VERDICT r4 weak spot 2): every earlier pin used ONE synthetic weight draw (salt 0) whose residual stream stays far below the
# |v| = 512 where the e4m3 lo plane of an fp16-rounded operand clamps (include/panacea_hip.h).  Two more pins of the reference's own
# forward (oracle/gen_golden_full.py --t 500 --wsalt 1 / --wtail 16): a second weight draw, and a heavy-tailed weight set whose
# "massive activation" channels (synth.synth_tensor: output rows c % 64 == 5 of every residual-out tensor x 64) carry the stream
# into the clamp range (|v| 512 .. ~2000; a first attempt with gain 256 put the stream at 2.3e4 and an fp16 operand beyond 65504:
# non-finite eps — an fp16 path's own range limit, stated in UNetModel3D.eps_contract).  Measured on the MI355X at |stream| = 1.8e3:
# 2.3e-3 max / 3.3e-4 mean — OUTSIDE the 1e-3 contract, whose range eps_contract therefore states (|stream| < 512: the e4m3 lo plane's
# range; the massive channels reach the stream-class convs at fp16 precision).  TAIL_GATE pins that measurement (x 1.3).
TAIL_GATE = 3.0e-3


@pytest.mark.parametrize("fname,wsalt,wtail", [("full_cfg3_t500_w1.npz", 1, 0.0), ("full_cfg3_t500_tail64.npz", 0, 64.0)])
def test_full_size_other_weight_sets(full_net, fname, wsalt, wtail):
    from panacea_amd import synth
    path = GOLDEN / fname
    if not path.exists():
        pytest.skip(f"{fname} not generated")
    w, kw = full_net
    gp = np.load(path)
    assert int(gp["weight_salt"]) == wsalt and float(gp["weight_tail"]) == wtail and int(gp["t_index"]) == 500
    try:
        w.diffusion_model.load_state_dict(synth.synth_state_dict(manifest("full"), salt=wsalt, tail=wtail), strict=True)
        gi = {k: v.to(DEV) for k, v in synth.synth_inputs(2, 8, 32, 384, context_dim=kw["context_dim"], t_index=500).items()}
        trace = {} if wtail else None
        eps = w(gi["x"], gi["t"], cond(gi), trace=trace) if wtail else w(gi["x"], gi["t"], cond(gi))
        st = err_stats(eps.reshape(-1)[::7], gp["eps_s7"])
        extra = {}
        if wtail:
            extra["stream_max_abs"] = max(float(v.abs().max()) for k, v in trace.items() if "blocks" in k or "middle" in k)
            del trace
        print(f"config 3, t=500, weight salt {wsalt}, tail {wtail} vs reference:", st, extra)
        # round 6 range monitor: the heavy-tail evaluation is DETECTED at run time (e4m3 lo planes saturated), the ordinary draw is not
        extra["lo_clamped"] = w.diffusion_model.lo_clamped
        measured("full_cfg3_weights", wsalt=wsalt, wtail=wtail, max_abs=st["max_abs"], mean_abs=st["mean_abs"], **extra)
        assert (extra["lo_clamped"] > 0) == bool(wtail), extra
        if wtail:
            assert extra["stream_max_abs"] >= 512.0, extra          # the regime the pin exists for: the e4m3 clamp range is reached
            w.diffusion_model.on_range_exceeded = "raise"
            try:
                with pytest.raises(RuntimeError, match="saturated"):
                    w(gi["x"], gi["t"], cond(gi))
                    w.diffusion_model.lo_clamped                     # (the count arrives asynchronously: asking for it waits)
            finally:
                w.diffusion_model.on_range_exceeded = "warn"
        assert st["max_abs"] <= (TAIL_GATE if wtail else NORTH_STAR), st
        if wtail:
            c = w.diffusion_model.eps_contract
            assert c["beyond"]["eps_max_abs_measured"] >= 0.75 * st["max_abs"], (c, st)      # the stated number is the measured one
    finally:
        w.diffusion_model.load_state_dict(synth.synth_state_dict(manifest("full")), strict=True)     # the module fixture's weights


def test_streams_and_graph_replay_are_bit_identical():
    """ControlNet on a side stream, per-sample stream pairs and hipGraph replay of a whole sampler step must not
    change a single bit (tiny network; the same invariants hold at full size because no kernel depends on the
    batch composition or on launch order)."""
    from panacea_amd import sampling as S
    from panacea_amd.graph import GraphedStep
    w, _, kw = product_network("tiny", DEV)
    net = w.diffusion_model
    inp = step_inputs("tiny", kw, DEV)
    net.two_stream, net.split_samples = False, False
    ref = w(inp["x"], inp["t"], cond(inp))
    net.two_stream = True
    assert torch.equal(w(inp["x"], inp["t"], cond(inp)), ref)
    net.split_samples = True
    assert torch.equal(w(inp["x"], inp["t"], cond(inp)), ref)
    net.split_samples = False
    # one Euler/CFG step, eager vs replayed from a captured graph (twice, with different latents)
    T = kw["num_frames"]
    c = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
    uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
    den = S.DiscreteDenoiser().to(DEV)
    smp = S.EulerEDMSampler(3, guider=S.VanillaCFG(5.0), device=DEV)
    sig = smp.sigmas()
    s_in = inp["x"].new_ones([T])
    step = lambda xi, s0, s1: smp.sampler_step(s0, s1, lambda a, b, cc: den(w, a, b, cc), xi, c, uc)   # noqa: E731
    x0 = inp["x"][T:] * 14.6
    with torch.no_grad():
        e0 = step(x0, s_in * sig[0], s_in * sig[1])
        e1 = step(e0, s_in * sig[1], s_in * sig[2])
        g = GraphedStep(step, x0, s_in * sig[0], s_in * sig[1])
        g0 = g(x0, s_in * sig[0], s_in * sig[1]).clone()
        g1 = g(g0, s_in * sig[1], s_in * sig[2]).clone()
    torch.cuda.synchronize()
    assert torch.equal(g0, e0) and torch.equal(g1, e1)
    # hoisted step invariants (text K/V + hint stem computed once per schedule): same bits, eager and graphed
    with torch.no_grad():
        c2, u2 = S.hoist_invariants(w, smp.guider, c, uc)
        hstep = lambda xi, s0, s1: smp.sampler_step(s0, s1, lambda a, b, cc: den(w, a, b, cc), xi, c2, u2)   # noqa: E731
        h0 = hstep(x0, s_in * sig[0], s_in * sig[1])
        h1 = hstep(h0, s_in * sig[1], s_in * sig[2])
        gh = GraphedStep(hstep, x0, s_in * sig[0], s_in * sig[1])
        gh1 = gh(gh(x0, s_in * sig[0], s_in * sig[1]).clone(), s_in * sig[1], s_in * sig[2]).clone()
    torch.cuda.synchronize()
    assert torch.equal(h0, e0) and torch.equal(h1, e1) and torch.equal(gh1, e1)


def _rccl_world1_worker(rank, port, out):
    """frame-shard exchanges through torch.distributed "nccl" (= RCCL) on the MI355X with a one-rank group"""
    import os
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    from helpers import cond as cond_of, product_network, step_inputs
    from panacea_amd import configs as cfgs, engine as E, parallel
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    grp = dist.new_group([0])
    kw = cfgs.with_frames(cfgs.get("tiny"), 4)
    w, _, _ = product_network("tiny", "cuda", kw=kw)
    inp = step_inputs("tiny", kw, "cuda", t_index=500, shape=(2, 4, 8, 96))
    ref = w(inp["x"], inp["t"], cond_of(inp))
    sh0 = E.FrameShard(1, 0, None)      # the loop-back without a group: the expected bits of the sharded form
    parallel.apply_frame_shard(w, sh0)
    ref_sharded = w(inp["x"], inp["t"], cond_of(inp))
    sh = E.FrameShard(1, 0, None)
    sh.group = grp                      # G = 1 over a real RCCL group: all_to_all_single / all_reduce / all_gather run on the GPU
    parallel.apply_frame_shard(w, sh)
    got = w(inp["x"], inp["t"], cond_of(inp))
    from helpers import gn_statistics_from_launches
    sht = E.FrameShard(1, 0, None, resblock="transpose")
    sht.group = grp                     # round 2's form of the ResBlock sites moves data only: the unsharded bits
    with gn_statistics_from_launches():
        parallel.apply_frame_shard(w, None)
        ref_l = w(inp["x"], inp["t"], cond_of(inp))
        parallel.apply_frame_shard(w, sht)
        got_t = w(inp["x"], inp["t"], cond_of(inp))
    torch.cuda.synchronize()
    ok = bool(torch.equal(got, ref_sharded)) and bool(torch.equal(got_t, ref_l)) and sh.exchanges > 0 \
        and (got - ref).abs().max().item() <= 1.2e-3
    open(out, "w").write("ok" if ok else f"mismatch {(got - ref_sharded).abs().max().item()} {(got_t - ref_l).abs().max().item()} "
                                        f"{(got - ref).abs().max().item()}")
    dist.destroy_process_group()


def test_frame_shard_code_path_single_device(tmp_path):
    """SURVEY §8e on ONE GPU: (1) the loop-back FrameShard (G = 1) drives every exchange site of the network — to_pixels /
    to_frames around the ResBlock3D temporal sites and the STT temporal branch, the embedding-row gather — and must not
    change a bit; (2) the same through a one-rank RCCL group in a child process (collectives on the HIP stream)."""
    from panacea_amd import engine as E, parallel
    kw = configs.with_frames(configs.get("tiny"), 4)
    w, _, _ = product_network("tiny", DEV, kw=kw)
    inp = step_inputs("tiny", kw, DEV, t_index=500, shape=(2, 4, 8, 96))
    ref = w(inp["x"], inp["t"], cond(inp))
    from helpers import gn_statistics_from_launches
    sh = E.FrameShard(1, 0, None, resblock="transpose")      # round 2's form: the exchanges only move data
    with gn_statistics_from_launches():                      # (every GroupNorm with its own statistics launch on both sides)
        ref_l = w(inp["x"], inp["t"], cond(inp))
        parallel.apply_frame_shard(w, sh)
        got = w(inp["x"], inp["t"], cond(inp))
    torch.cuda.synchronize()
    assert sh.exchanges >= 20 and torch.equal(got, ref_l) and (ref_l - ref).abs().max().item() <= 1.2e-3
    # round 4's form of the ResBlock3D temporal sites (partial sums + halo frames, the fp32 stream stays put): other roundings of
    # the temporal GroupNorm statistics, so eps differs like two `precise` evaluations do
    sh = E.FrameShard(1, 0, None)
    parallel.apply_frame_shard(w, sh)
    got = w(inp["x"], inp["t"], cond(inp))
    torch.cuda.synchronize()
    d = (got - ref).abs()
    print(f"frame loop-back (halo form) vs unsharded: max {d.max().item():.3e} mean {d.mean().item():.3e}; {sh.exchanges} exchanges")
    assert sh.exchanges >= 20 and d.max().item() <= 1.2e-3 and d.mean().item() <= 2e-4
    import torch.multiprocessing as mp
    out = tmp_path / "rccl.txt"
    mp.get_context("spawn")
    mp.spawn(_rccl_world1_worker, args=(29400 + (hash(str(tmp_path)) % 500), str(out)), nprocs=1, join=True)
    assert out.read_text() == "ok", out.read_text()


def test_fused_sampler_step_on_device_replays_the_reference_trajectory():
    """SURVEY §8 f1 on the GPU: (1) pnc_cfg_euler_step replays the reference's 3- and 25-step Euler/CFG trajectory
    (tests/golden/sampler.npz, produced by the reference's own sampler / denoiser / guider classes); (2) the fused
    sampler step of the product (BoundDenoiser: c_in + CFG doubling in the entry kernel, one exit kernel) gives the bits
    of the plain step, eagerly and replayed from a hipGraph."""
    import test_sampling as ts
    from panacea_amd import sampling as S
    from panacea_amd.graph import GraphedStep
    for n in (3, 25):
        xs = ts._trajectory_through_exit_kernel(hip, n, DEV)
        assert np.allclose(xs.cpu().numpy(), ts.G[f"sampler.{n}.x_final"], atol=2e-5, rtol=1e-5)
    w, _, kw = product_network("tiny", DEV)
    inp = step_inputs("tiny", kw, DEV)
    T = kw["num_frames"]
    c = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
    uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
    den = S.DiscreteDenoiser().to(DEV)
    smp = S.EulerEDMSampler(3, guider=S.VanillaCFG(5.0), device=DEV)
    sig = smp.sigmas()
    x0 = inp["x"][T:] * 14.6
    s_in = x0.new_ones([T])
    bd = S.BoundDenoiser(den, w)
    with torch.no_grad():
        assert smp._fusable(bd, x0, c)
        fused = smp.sampler_step(s_in * sig[0], s_in * sig[1], bd, x0, c, uc)
        smp.fuse = False
        plain = smp.sampler_step(s_in * sig[0], s_in * sig[1], bd, x0, c, uc)
        smp.fuse = True
        g = GraphedStep(lambda xi, s0, s1: smp.sampler_step(s0, s1, bd, xi, c, uc), x0, s_in * sig[0], s_in * sig[1])
        graphed = g(x0, s_in * sig[0], s_in * sig[1]).clone()
    torch.cuda.synchronize()
    # same arithmetic, in the reference's rounding order, as the torch elementwise kernels of the plain step (no FMA
    # contraction in the exit kernel): bit-equal where torch's division is IEEE, else within an ulp of |x| ~ 50
    print("fused == plain bitwise:", torch.equal(fused, plain), (fused - plain).abs().max().item())
    assert torch.allclose(fused, plain, rtol=0, atol=2e-5)
    assert torch.equal(graphed, fused)
    # the BEV hint as ONE tensor shared by c and uc (what the reference's conditioner hands over): the hint stem then runs on
    # T frames instead of 2 T and nothing is concatenated — same bits as the doubled evaluation above
    assert torch.equal(uc["cond_feat"], c["cond_feat"])
    uc_shared = dict(uc, cond_feat=c["cond_feat"])
    with torch.no_grad():
        shared = smp.sampler_step(s_in * sig[0], s_in * sig[1], bd, x0, c, uc_shared)
    torch.cuda.synchronize()
    assert torch.equal(shared, fused)


def test_yaml_exact_25_step_trajectory_vs_oracle_on_the_gpu():
    """BASELINE config 5's setup (25-step schedule, last-frame `concat` conditioning, share-noise init) through the HIP
    path with hoisted invariants and the fused device step, against the sampler mirrors driven by the CPU oracle."""
    import test_sampling as ts
    ref, got = ts._yaml_exact_trajectories(DEV)
    rms = ref.pow(2).mean().sqrt().item()
    d = (got - ref).abs()
    print("config-5 setup, 25 steps:", d.max().item(), d.mean().item(), "latent rms", rms)
    assert d.max().item() <= 3e-3 * rms and d.mean().item() <= 5e-4 * rms
