"""N > 1 path on CPU: world_size-2 gloo processes.  (1) CFG-half sharding (`ShardedCFG`, one all-gather per step)
reproduces the single-process VanillaCFG trajectory; (2) replica mode: ranks run independent samples and the
bench-style max-over-ranks timing reduction works.  The host logic runs against the torch emulation of the
C-ABI (tests/emu.py), as in tests/test_engine_emu.py."""
import os
import sys
import tempfile
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import emu
    from helpers import product_network, step_inputs
    from panacea_amd import engine as E, parallel, sampling as S
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    groups = parallel.cfg_pair_groups(world)
    net, _, kw = product_network("tiny")
    inp = step_inputs("tiny", kw)
    T = kw["num_frames"]
    cond = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
    uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
    den = S.DiscreteDenoiser()
    x0 = inp["x"][T:].clone()
    with E.use_backend(emu), torch.no_grad():
        denoiser = lambda xi, sigma, cc: den(net, xi, sigma, cc)     # noqa: E731
        smp = S.EulerEDMSampler(3, guider=parallel.ShardedCFG(5.0, groups[rank // 2], rank % 2), device="cpu")
        xs = smp(denoiser, x0.clone(), cond, uc)
        # hoisted step invariants compose with the sharded guider (each rank prepares its own half): same bits
        xs_h = smp(denoiser, x0.clone(), cond, uc, network=net)
        assert torch.equal(xs_h, xs)
        # the same sharding around a closed-form network: must agree with the single-process guider exactly
        def fake_net(a, t, c_):       # per-sample closed form (text statistic of the frame's own sample)
            txt = c_["crossattn"].mean(dim=(1, 2)).repeat_interleave(a.shape[0] // c_["crossattn"].shape[0])
            return torch.tanh(0.3 * a) * 0.5 + 1e-4 * t.float()[:, None, None, None] + txt[:, None, None, None] \
                + 0.05 * c_["concat"]
        fake = lambda xi, sigma, cc: den(fake_net, xi, sigma, cc)     # noqa: E731
        xs_f = smp(fake, x0.clone(), cond, uc)
        if rank == 0:
            single = S.EulerEDMSampler(3, guider=S.VanillaCFG(5.0), device="cpu")
            torch.save({"sharded": xs, "single": single(denoiser, x0.clone(), cond, uc),
                        "sharded_fake": xs_f, "single_fake": single(fake, x0.clone(), cond, uc)}, Path(out_dir) / "cfg.pt")
    # both ranks of a pair hold the same next latent (no further exchange needed)
    both = [torch.empty_like(xs), torch.empty_like(xs)]
    dist.all_gather(both, xs)
    assert torch.equal(both[0], both[1])
    # bench-style reduction: max over ranks of a per-rank elapsed time
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    assert parallel.replica_seed(rank) == 3407 + rank
    # a pair that was seeded per RANK is refused instead of silently combining two different latents
    g = parallel.ShardedCFG(5.0, groups[rank // 2], rank % 2)
    g.check_pair_consistency(x0, torch.ones(T))
    try:
        g.check_pair_consistency(x0 + rank, torch.ones(T))
        raise AssertionError("a per-rank latent must be refused")
    except RuntimeError:
        pass
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_cfg_sharding_and_replicas_world2():
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        r = torch.load(Path(d) / "cfg.pt")
    assert torch.allclose(r["sharded_fake"], r["single_fake"], rtol=0, atol=1e-5 * r["single_fake"].abs().max().item())
    # real network: the torch emulation is not batch-invariant (B=1 vs B=2 matmuls round differently), and a 1e-7
    # difference decorrelates the fp16 operand rounding downstream; the HIP kernels ARE batch-
    # invariant and this equality is asserted bit-exactly on the GPU (tests/test_model_gpu.py).
    err = (r["sharded"] - r["single"]).abs().max().item()
    assert err <= 5e-3 * r["single"].abs().max().item(), err


# ------------------------------------------------------------------------ frame-group sharding
def _frames_worker(rank, world, port, out_dir, cfg, G, T, resblock="halo"):
    """rank grid cfg x G over one sample of the tiny network with T frames: one eps evaluation and a 3-step schedule"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import emu
    from helpers import cond as cond_of, product_network, step_inputs
    from panacea_amd import configs, engine as E, parallel, sampling as S
    parallel.init_distributed("gloo")
    lo = parallel.RankLayout(world, rank, cfg=cfg, frames=G)
    assert lo.samples == 1 and lo.name
    groups = parallel.Groups(lo)
    shard = groups.frame_shard()
    shard.resblock = resblock
    kw = configs.with_frames(configs.get("tiny"), T)
    net, _, _ = product_network("tiny", kw=kw)
    inp = step_inputs("tiny", kw, t_index=500, shape=(2, T, 8, 96))
    # --- (0) round 4: partial sums added over the frame group, halo frames from the neighbour ranks (zeros at the clip's ends)
    tl0 = T // G
    clip = torch.arange(2 * T * 6 * 4, dtype=torch.float32).view(2, T, 6, 4) + 1.0
    pad16 = torch.zeros(2, tl0 + 2, 6, 4, dtype=torch.float16)
    pad8 = torch.zeros(2, tl0 + 2, 6, 4, dtype=torch.uint8)
    pad16[:, 1:tl0 + 1] = clip[:, lo.frame_group * tl0:(lo.frame_group + 1) * tl0].half()
    pad8[:, 1:tl0 + 1] = (clip[:, lo.frame_group * tl0:(lo.frame_group + 1) * tl0] % 251).to(torch.uint8)
    pad16[:, 0] = pad16[:, -1] = 7.0                     # stale values the exchange must overwrite
    shard.halo_frames([pad16, pad8], 2, tl0)
    want = torch.zeros(2, T + 2, 6, 4)
    want[:, 1:T + 1] = clip
    lo_f, hi_f = lo.frame_group * tl0, (lo.frame_group + 1) * tl0
    assert torch.equal(pad16.float(), want[:, lo_f:hi_f + 2].half().float())
    assert torch.equal(pad8, (want[:, lo_f:hi_f + 2] % 251).to(torch.uint8) * (want[:, lo_f:hi_f + 2] > 0))
    part = torch.full((5,), float(lo.frame_group + 1))
    assert torch.equal(shard.allreduce_sum(part), torch.full((5,), float(G * (G + 1) // 2)))
    # --- (1) the exchanges themselves: to_pixels / to_frames against the global tensor
    Bx, N, C = 2, 48, 8
    full = torch.arange(Bx * T * N * C, dtype=torch.float32).view(Bx, T, N, C)
    tl, Np = T // G, N // G
    mine = full[:, lo.frame_group * tl:(lo.frame_group + 1) * tl].reshape(-1, C)
    px = shard.to_pixels(mine, Bx, N)
    assert torch.equal(px.view(Bx, T, Np, C), full[:, :, lo.frame_group * Np:(lo.frame_group + 1) * Np])
    assert torch.equal(shard.to_frames(px, Bx, N), mine)
    rows = shard.gather_rows(full[:, lo.frame_group * tl:(lo.frame_group + 1) * tl, 0, :].reshape(-1, C), Bx)
    assert torch.equal(rows, full[:, :, 0, :].reshape(-1, C))
    # --- (2) one network evaluation on this rank's frames of this rank's half/halves
    halves = [lo.half] if cfg == 2 else [0, 1]
    def pick(t):                                   # per-frame tensor of the CFG batch -> this rank's rows
        v = t.view(2, T, *t.shape[1:])[halves]
        return v[:, lo.frame_group * tl:(lo.frame_group + 1) * tl].reshape(-1, *t.shape[1:]).contiguous()
    loc = {"x": pick(inp["x"]), "t": pick(inp["t"]), "concat": pick(inp["concat"]), "cond_feat": pick(inp["cond_feat"]),
           "crossattn": inp["crossattn"][halves]}
    parallel.apply_frame_shard(net, shard)
    with E.use_backend(emu), torch.no_grad():
        eps_loc = net(loc["x"], loc["t"], cond_of(loc))
        assert shard.exchanges > 0 and shard.bytes_sent > 0
        torch.save({"eps": eps_loc, "halves": halves, "fg": lo.frame_group}, Path(out_dir) / f"eps{rank}.pt")
        # --- (3) three sampler steps with the layout's guider; the latent stays sharded, gathered once at the end
        cond = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
        uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
        den = S.DiscreteDenoiser()
        denoiser = lambda xi, sigma, cc: den(net, xi, sigma, cc)     # noqa: E731
        smp = S.EulerEDMSampler(3, guider=groups.guider(5.0), device="cpu")
        x0 = inp["x"][T:]
        xs = smp(denoiser, parallel.local_frames(x0, lo, T), parallel.shard_conditioning(cond, lo, T),
                 parallel.shard_conditioning(uc, lo, T))
        xs_all = parallel.gather_frames(xs, groups, T)
        # fused sampler tail on the frame-sharded network (cfg = 1: both halves on the rank; cfg = 2: the pair gathers its eps)
        x_l = parallel.local_frames(x0, lo, T)
        c_l, u_l = parallel.shard_conditioning(cond, lo, T), parallel.shard_conditioning(uc, lo, T)
        sig = smp.sigmas()
        s_in = x_l.new_ones([x_l.shape[0]])
        xin = x_l * torch.sqrt(1.0 + sig[0] ** 2.0)
        plain = smp.sampler_step(s_in * sig[0], s_in * sig[1], denoiser, xin, c_l, u_l)
        fused = smp._fused_step(s_in * sig[0], s_in * sig[1], S.BoundDenoiser(den, net), xin, c_l, u_l)
        assert torch.equal(plain, fused), (plain - fused).abs().max().item()
        if rank == 0:
            parallel.apply_frame_shard(net, None)
            eps_ref = net(inp["x"], inp["t"], cond_of(inp))
            single = S.EulerEDMSampler(3, guider=S.VanillaCFG(5.0), device="cpu")
            torch.save({"eps_ref": eps_ref, "traj": xs_all, "traj_ref": single(denoiser, x0.clone(), cond, uc)},
                       Path(out_dir) / "ref.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,cfg,G,T,resblock", [(2, 1, 2, 4, "halo"), (4, 2, 2, 4, "halo"), (4, 1, 4, 4, "halo"), (2, 1, 2, 4, "transpose")])
def test_frame_group_sharding_reproduces_the_single_process_eps(world, cfg, G, T, resblock):
    """Frames of a sample over G ranks (x CFG halves): eps of every rank's frames and a 3-step trajectory equal the
    single-process result.  Tolerance: the torch emulation is not batch-invariant (GEMMs over M/G rows round differently)
    and a 1e-7 difference decorrelates the fp16 operand rounding downstream — the bound is the stated eps tolerance
    (1e-3 max-abs, BASELINE.json north_star), the observed difference is ~1e-4."""
    from panacea_amd.parallel import RankLayout
    port = 29500 + ((os.getpid() * 7 + world * 13 + G + len(resblock)) % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_frames_worker, args=(world, port, d, cfg, G, T, resblock), nprocs=world, join=True)
        ref = torch.load(Path(d) / "ref.pt")
        eps_ref = ref["eps_ref"].view(2, T, *ref["eps_ref"].shape[1:])
        tl = T // G
        worst = 0.0
        for r in range(world):
            e = torch.load(Path(d) / f"eps{r}.pt")
            want = eps_ref[e["halves"]][:, e["fg"] * tl:(e["fg"] + 1) * tl].reshape(e["eps"].shape)
            worst = max(worst, (e["eps"] - want).abs().max().item())
        print(f"world {world} cfg {cfg} G {G}: max |eps_sharded - eps_single| = {worst:.3e}")
        assert worst <= 1e-3
        err = (ref["traj"] - ref["traj_ref"]).abs().max().item()
        assert err <= 5e-3 * ref["traj_ref"].abs().max().item(), err
    lo = RankLayout(8, 5, cfg=2, frames=4)
    assert (lo.sample, lo.half, lo.frame_group) == (0, 1, 1) and lo.frame_group_ranks(0, 1) == [4, 5, 6, 7]
    assert lo.cfg_pair_ranks(0, 1) == [1, 5]


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from panacea_amd import parallel
    parallel.init_distributed("gloo")
    layout = parallel.RankLayout(world, rank, cfg=1, frames=2)
    groups = parallel.Groups(layout)
    B, T = 3, 4                                            # THREE samples per rank: a plain cat of the parts interleaves them
    full = (torch.arange(B * T, dtype=torch.float32)[:, None] * 10 + torch.arange(5, dtype=torch.float32)[None]).view(B * T, 5)
    mine = parallel.local_frames(full, layout, T)
    assert mine.shape[0] == B * T // 2
    back = parallel.gather_frames(mine, groups, T)
    ok = torch.equal(back, full)
    (Path(out_dir) / f"gather{rank}.txt").write_text("ok" if ok else f"order {back[:, 0].tolist()}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gather_frames_restores_sample_major_order_for_several_samples_per_rank():
    """gather_frames() must return rows in (sample, frame) order — the order local_frames() cut them in — also when a
    rank carries more than one sample"""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gather_worker, args=(2, 29611 + os.getpid() % 200, d), nprocs=2, join=True)
        assert [(Path(d) / f"gather{r}.txt").read_text() for r in range(2)] == ["ok", "ok"]


# ------------------------------------------------------------------------ view-group sharding
def _views_worker(rank, world, port, out_dir, cfg, V, T, network):
    """rank grid cfg x V over one sample: the exchanges against the global tensors, then (network=True) one eps evaluation and a
    2-step schedule of the tiny network on this rank's band of views"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import emu
    import torch.nn.functional as TF
    from helpers import cond as cond_of, product_network, step_inputs
    from panacea_amd import configs, engine as E, parallel, sampling as S
    from panacea_amd.nn.attention import INTER_SEGS
    parallel.init_distributed("gloo")
    lo = parallel.RankLayout(world, rank, cfg=cfg, views=V)
    assert lo.samples == 1 and "views" in lo.name
    groups = parallel.Groups(lo)
    vs = groups.view_shard()
    vg, nl = lo.view_group, 6 // V
    assert (vs.n_local, vs.first) == (nl, vg * nl)
    # --- (1a) 3x3 convs over the band + halo equal the band of the conv over the panorama (stride 1, stride 2, nearest x2)
    g = torch.Generator().manual_seed(5)
    Fx, H, W, C = 2, 4, 24, 8
    full = torch.randn(Fx, H, W, C, generator=g)
    wgt = torch.randn(5, C, 3, 3, generator=g)
    wl = W // V
    mine = full[:, :, vg * wl:(vg + 1) * wl].contiguous()
    import types
    from panacea_amd import engine as E
    stub = types.SimpleNamespace(device=torch.device("cpu"))
    stub.empty = lambda shape, dtype, tail_rows=0: E.Runtime.empty(stub, shape, dtype, tail_rows)
    full, wgt = full.half().float(), wgt.half().float()
    w16 = E.pk_conv3x3(wgt)
    mine = full[:, :, vg * wl:(vg + 1) * wl].contiguous()
    for stride, up in ((1, False), (2, False), (1, True)):
        for in_place in (False, True):
            if in_place:        # an operand that was allocated with room for the columns is used where it lies
                op = stub.empty((Fx * H * wl, C), torch.float16, 2 * Fx * H)
                op.copy_(mine.view(-1, C))
            else:
                op = mine.half().view(-1, C)
            (plane,), xoff = vs.band_operand(stub, [op], Fx, H, wl, C)
            assert (plane.data_ptr() == op.data_ptr()) == in_place and xoff == Fx * H * wl * C
            Hout = 2 * H if up else (H - 1) // stride + 1
            Wout = 2 * wl if up else (wl - 1) // stride + 1
            got = torch.empty(Fx * Hout * Wout, 5)
            emu.gemm(plane, w16, M=got.shape[0], N=5, K=9 * C, a_mode=emu.A_CONV3X3,
                     conv=dict(Cin=C, Hin=H, Win=wl, Hout=Hout, Wout=Wout, stride=stride, upsample=int(up), x_halo_off=xoff),
                     out32=got, ldc32=5)

            def conv(t):
                t = t.permute(0, 3, 1, 2)
                if up:
                    t = TF.interpolate(t, scale_factor=2, mode="nearest")
                return TF.conv2d(t, wgt, stride=stride, padding=1)
            want = conv(full)[..., vg * Wout:(vg + 1) * Wout].permute(0, 2, 3, 1).reshape(-1, 5)
            assert torch.allclose(got, want, atol=1e-4), (stride, up, in_place, (got - want).abs().max())
    # --- (1b) GroupNorm statistics of the whole panorama from the bands' records
    x32 = torch.randn(Fx * H * wl, 64, generator=torch.Generator().manual_seed(11 + vg)) * (1 + vg) + vg
    ppc = 8
    nchunk = (H * wl + ppc - 1) // ppc
    part = torch.empty(Fx * nchunk * 32 * 3)
    emu.groupnorm_stats(x32, 64, Fx, H * wl, 64, ppc, part)
    rec = vs.combine_stats(part, Fx, nchunk, emu).view(Fx, nchunk, 32, 3)
    allx = [torch.empty_like(x32) for _ in range(V)]
    dist.all_gather(allx, x32, group=groups.view_group)
    pano = torch.cat([a.view(Fx, H * wl, 32, 2) for a in allx], dim=1).permute(0, 2, 1, 3).reshape(Fx, 32, -1).double()
    assert torch.allclose(rec[:, 0, :, 1].double(), pano.mean(-1), atol=1e-5)
    assert torch.allclose(rec[:, 0, :, 2].double() / rec[:, 0, :, 0].double(), pano.var(-1, unbiased=False), rtol=1e-4)
    assert nchunk > 1 and not rec[:, 1:].any()            # one combined record per frame, the other slots empty
    assert torch.allclose(rec[:, 0, :, 0], torch.full((Fx, 32), float(V * H * wl * 2)))
    # --- (1c) neighbour views + local segments: every local view finds exactly the views INTER_SEGS names
    tag = torch.arange(6, dtype=torch.float32).repeat_interleave(W // 6)                 # view id per panorama column
    k4 = tag[vg * wl:(vg + 1) * wl].view(1, 1, wl, 1).expand(1, 2, wl, 3).contiguous()
    v4 = k4.permute(0, 3, 1, 2).contiguous()
    ke, ve = vs.neighbour_views(k4, v4)
    wv = wl // nl
    assert ke.shape == (1, 2, wl + 2 * wv, 3) and ve.shape == (1, 3, 2, wl + 2 * wv)
    for i, row in enumerate(vs.local_segments(INTER_SEGS)):
        assert [int(ke[0, 0, u * wv, 0]) for u in row] == INTER_SEGS[vs.first + i]
        assert [int(ve[0, 0, 0, u * wv]) for u in row] == INTER_SEGS[vs.first + i]
    assert torch.equal(vs.gather_width(tag[None, vg * wl:(vg + 1) * wl])[0], tag)
    if not network:
        dist.barrier()
        dist.destroy_process_group()
        return
    # --- (2) one network evaluation on this rank's band of this rank's half/halves
    kw = configs.with_frames(configs.get("tiny"), T)
    net, _, _ = product_network("tiny", kw=kw)
    inp = step_inputs("tiny", kw, t_index=500, shape=(2, T, 8, 96))
    halves = [lo.half] if cfg == 2 else [0, 1]

    def pick(t, per_frame=True):
        v = t.view(2, T, *t.shape[1:])[halves].reshape(-1, *t.shape[1:])
        return parallel.local_views(v, lo) if v.dim() == 4 else v.contiguous()
    loc = {"x": pick(inp["x"]), "t": pick(inp["t"]), "concat": pick(inp["concat"]), "cond_feat": pick(inp["cond_feat"]),
           "crossattn": inp["crossattn"][halves]}
    parallel.apply_view_shard(net, vs)
    with E.use_backend(emu), torch.no_grad():
        # round 6: GroupNorm -> 3x3 conv sites exchange their records and the neighbours' RAW edge columns in ONE all-to-all
        # (ViewShard.stats_and_halo) instead of records, then normalised columns: fewer exchanges, the same bits
        vs.fused_halo = False
        n0 = vs.exchanges
        eps_two = net(loc["x"], loc["t"], cond_of(loc))
        n_two = vs.exchanges - n0
        vs.fused_halo = True
        n0 = vs.exchanges
        eps_loc = net(loc["x"], loc["t"], cond_of(loc))
        assert torch.equal(eps_loc, eps_two), (eps_loc - eps_two).abs().max().item()
        assert vs.exchanges - n0 <= n_two - 20, (vs.exchanges - n0, n_two)         # the tiny network: 23 GroupNorm + conv sites
        assert vs.exchanges > n0 and vs.bytes_sent > 0
        torch.save({"eps": eps_loc, "halves": halves, "vg": vg, "exchanges": vs.exchanges - n0}, Path(out_dir) / f"eps{rank}.pt")
        # --- (3) two sampler steps with the layout's guider; the latent stays a band, gathered once at the end
        cond = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
        uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
        den = S.DiscreteDenoiser()
        denoiser = lambda xi, sigma, cc: den(net, xi, sigma, cc)     # noqa: E731
        smp = S.EulerEDMSampler(2, guider=groups.guider(5.0), device="cpu")
        x0 = inp["x"][T:]
        xs = smp(denoiser, parallel.local_views(x0, lo), parallel.shard_conditioning(cond, lo, T),
                 parallel.shard_conditioning(uc, lo, T))
        xs_all = parallel.gather_views(xs, groups)
        # --- (4) the same schedule with the step invariants hoisted (hint stem on the band — its convs exchange halos once —, text
        #         K/V): bit-identical to the schedule that recomputes them every step
        n1 = vs.exchanges
        ch, uh = S.hoist_invariants(net, smp.guider, parallel.shard_conditioning(cond, lo, T), parallel.shard_conditioning(uc, lo, T))
        xs_h = smp(denoiser, parallel.local_views(x0, lo), ch, uh)
        assert torch.equal(xs_h, xs), (xs_h - xs).abs().max().item()
        assert vs.exchanges - n1 < 2 * (n1 - n0)        # fewer exchanges per step: the hint stem's halos are not repeated
        if rank == 0:
            parallel.apply_view_shard(net, None)
            eps_ref = net(inp["x"], inp["t"], cond_of(inp))
            single = S.EulerEDMSampler(2, guider=S.VanillaCFG(5.0), device="cpu")
            torch.save({"eps_ref": eps_ref, "traj": xs_all, "traj_ref": single(denoiser, x0.clone(), cond, uc)},
                       Path(out_dir) / "ref.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_view_group_exchanges_with_one_view_per_rank():
    """six ranks, one view each: halos, panorama statistics, neighbour views (the circular wrap 0 <-> 5 and the reference's
    one-sided view 5: attention.py:545-559)"""
    port = 29500 + ((os.getpid() * 11 + 77) % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_views_worker, args=(6, port, d, 1, 6, 2, False), nprocs=6, join=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,cfg,V,T", [(2, 1, 2, 2), (3, 1, 3, 2), (4, 2, 2, 2)])
def test_view_group_sharding_reproduces_the_single_process_eps(world, cfg, V, T):
    """The six views of a sample over V ranks (x CFG halves): eps of every rank's band and a 2-step trajectory equal the
    single-process result.  The exchanges are checked exactly in the worker (1a-1c); end to end the two evaluations differ by
    decorrelated fp16 operand roundings (torch's conv over a band sums in another order than over the panorama, the panorama
    statistics are combined from other partial records): measured 7.6-9.8e-4 max / 1.3e-4 mean, uniform over the columns of a
    band — no concentration at the band edges, where a halo or neighbour-view mistake would sit (|eps| <= 2.7)."""
    from panacea_amd.parallel import RankLayout, layout_for
    port = 29500 + ((os.getpid() * 7 + world * 17 + V) % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_views_worker, args=(world, port, d, cfg, V, T, True), nprocs=world, join=True)
        ref = torch.load(Path(d) / "ref.pt")
        eps_ref = ref["eps_ref"].view(2, T, *ref["eps_ref"].shape[1:])
        wl = eps_ref.shape[-1] // V
        worst = mean = 0.0
        for r in range(world):
            e = torch.load(Path(d) / f"eps{r}.pt")
            want = eps_ref[e["halves"]][..., e["vg"] * wl:(e["vg"] + 1) * wl].reshape(e["eps"].shape)
            diff = (e["eps"] - want).abs()
            worst, mean = max(worst, diff.max().item()), max(mean, diff.mean().item())
            edge = diff[..., [0, -1]].mean().item()                  # the columns next to a neighbour band
            assert edge <= 2.0 * diff.mean().item(), (edge, diff.mean().item())
        print(f"world {world} cfg {cfg} V {V}: |eps_sharded - eps_single| max {worst:.3e} mean {mean:.3e}, "
              f"{e['exchanges']} exchanges per evaluation")
        assert worst <= 2e-3 and mean <= 2.5e-4
        err = (ref["traj"] - ref["traj_ref"]).abs().max().item()
        assert err <= 5e-3 * ref["traj_ref"].abs().max().item(), err
    lo = RankLayout(12, 7, cfg=2, views=3)
    assert (lo.sample, lo.half, lo.view_group) == (1, 0, 1) and lo.view_group_ranks(1, 0) == [6, 7, 8]
    assert lo.cfg_pair_ranks(1, 1) == [7, 10]
    assert layout_for(4, 3, "cfg+views").views == 2 and layout_for(6, 0, "views").views == 6
    assert RankLayout(8, 0, cfg=2, frames=2, views=2).per_sample == 8          # round 4: the two splits compose


# ------------------------------------------------------------------------ cfg x view groups x frame groups (SURVEY §8e's 8-GPU grid)
def _grid_worker(rank, world, port, out_dir, cfg, G, V, T):
    """rank grid cfg x G frame groups x V view groups over ONE sample of the tiny network: every rank holds T/G frames of a band
    of W/V columns of its half; one eps evaluation and a 2-step schedule"""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import emu
    from helpers import cond as cond_of, product_network, step_inputs
    from panacea_amd import configs, engine as E, parallel, sampling as S
    parallel.init_distributed("gloo")
    lo = parallel.RankLayout(world, rank, cfg=cfg, frames=G, views=V)
    assert lo.samples == 1 and "frames" in lo.name and "views" in lo.name
    groups = parallel.Groups(lo)
    shard, vs = groups.frame_shard(), groups.view_shard()
    assert (shard.G, shard.index, vs.G, vs.index) == (G, lo.frame_group, V, lo.view_group)
    # the two groups of a rank are orthogonal: same (sample, half), one coordinate fixed each
    assert lo.frame_group_ranks(lo.sample, lo.half, lo.view_group)[lo.frame_group] == rank
    assert lo.view_group_ranks(lo.sample, lo.half, lo.frame_group)[lo.view_group] == rank
    kw = configs.with_frames(configs.get("tiny"), T)
    net, _, _ = product_network("tiny", kw=kw)
    inp = step_inputs("tiny", kw, t_index=500, shape=(2, T, 8, 96))
    halves = [lo.half] if cfg == 2 else [0, 1]
    tl = T // G

    def pick(t):
        v = t.view(2, T, *t.shape[1:])[halves][:, lo.frame_group * tl:(lo.frame_group + 1) * tl].reshape(-1, *t.shape[1:])
        return parallel.local_views(v, lo) if v.dim() == 4 else v.contiguous()
    loc = {"x": pick(inp["x"]), "t": pick(inp["t"]), "concat": pick(inp["concat"]), "cond_feat": pick(inp["cond_feat"]),
           "crossattn": inp["crossattn"][halves]}
    parallel.apply_frame_shard(net, shard)
    parallel.apply_view_shard(net, vs)
    with E.use_backend(emu), torch.no_grad():
        eps_loc = net(loc["x"], loc["t"], cond_of(loc))
        assert shard.exchanges > 0 and vs.exchanges > 0
        torch.save({"eps": eps_loc, "halves": halves, "fg": lo.frame_group, "vg": lo.view_group,
                    "frame_exchanges": shard.exchanges, "view_exchanges": vs.exchanges}, Path(out_dir) / f"eps{rank}.pt")
        cond = {"crossattn": inp["crossattn"][1:2], "concat": inp["concat"][T:], "cond_feat": inp["cond_feat"][T:]}
        uc = {"crossattn": inp["crossattn"][0:1], "concat": inp["concat"][:T], "cond_feat": inp["cond_feat"][:T]}
        den = S.DiscreteDenoiser()
        denoiser = lambda xi, sigma, cc: den(net, xi, sigma, cc)     # noqa: E731
        smp = S.EulerEDMSampler(2, guider=groups.guider(5.0), device="cpu")
        x0 = inp["x"][T:]
        xs = smp(denoiser, parallel.local_views(parallel.local_frames(x0, lo, T), lo), parallel.shard_conditioning(cond, lo, T),
                 parallel.shard_conditioning(uc, lo, T))
        xs_all = parallel.gather_frames(parallel.gather_views(xs, groups), groups, T)
        # the fused sampler tail in a sharded layout (round 4): entry + exit kernels on this rank's (frames, band) block, the
        # CFG pair's eps halves all-gathered in front of the exit kernel — same bits as the plain step of the same layout
        x_l = parallel.local_views(parallel.local_frames(x0, lo, T), lo)
        c_l, u_l = parallel.shard_conditioning(cond, lo, T), parallel.shard_conditioning(uc, lo, T)
        sig = smp.sigmas()
        s_in = x_l.new_ones([x_l.shape[0]])
        xin = x_l * torch.sqrt(1.0 + sig[0] ** 2.0)
        plain = smp.sampler_step(s_in * sig[0], s_in * sig[1], denoiser, xin, c_l, u_l)
        fused = smp._fused_step(s_in * sig[0], s_in * sig[1], S.BoundDenoiser(den, net), xin, c_l, u_l)
        assert torch.equal(plain, fused), (plain - fused).abs().max().item()
        if rank == 0:
            parallel.apply_frame_shard(net, None)
            parallel.apply_view_shard(net, None)
            eps_ref = net(inp["x"], inp["t"], cond_of(inp))
            single = S.EulerEDMSampler(2, guider=S.VanillaCFG(5.0), device="cpu")
            torch.save({"eps_ref": eps_ref, "traj": xs_all, "traj_ref": single(denoiser, x0.clone(), cond, uc)},
                       Path(out_dir) / "ref.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_cfg_x_views_x_frames_grid_world8_reproduces_the_single_process_eps():
    """SURVEY §8(e)'s 8-GPU partitioning — 2 CFG halves x 2 view groups (3 views each) x 2 frame groups — as eight gloo ranks
    on the tiny network (T = 4): every rank's (frames, band) block of eps and a 2-step trajectory equal the single-process
    result within the band the view layout alone shows (decorrelated fp16 operand roundings, tests above)."""
    from panacea_amd.parallel import RankLayout, layout_for
    world, cfg, G, V, T = 8, 2, 2, 2, 4
    port = 29500 + ((os.getpid() * 5 + 311) % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_grid_worker, args=(world, port, d, cfg, G, V, T), nprocs=world, join=True)
        ref = torch.load(Path(d) / "ref.pt")
        eps_ref = ref["eps_ref"].view(2, T, *ref["eps_ref"].shape[1:])
        tl, wl = T // G, eps_ref.shape[-1] // V
        worst = mean = 0.0
        for r in range(world):
            e = torch.load(Path(d) / f"eps{r}.pt")
            want = eps_ref[e["halves"]][:, e["fg"] * tl:(e["fg"] + 1) * tl, ..., e["vg"] * wl:(e["vg"] + 1) * wl].reshape(e["eps"].shape)
            diff = (e["eps"] - want).abs()
            worst, mean = max(worst, diff.max().item()), max(mean, diff.mean().item())
        print(f"world 8 = cfg 2 x frames 2 x views 2: |eps_sharded - eps_single| max {worst:.3e} mean {mean:.3e}; "
              f"{e['frame_exchanges']} frame + {e['view_exchanges']} view exchanges per evaluation")
        assert worst <= 2e-3 and mean <= 2.5e-4
        err = (ref["traj"] - ref["traj_ref"]).abs().max().item()
        assert err <= 5e-3 * ref["traj_ref"].abs().max().item(), err
    lo = RankLayout(8, 5, cfg=2, frames=2, views=2)
    assert (lo.sample, lo.half, lo.frame_group, lo.view_group) == (0, 1, 0, 1)
    assert lo.frame_group_ranks(0, 1, 1) == [5, 7] and lo.view_group_ranks(0, 1, 0) == [4, 5] and lo.cfg_pair_ranks(0, 1) == [1, 5]
    assert layout_for(8, 5, "cfg+views+frames") == lo
    assert layout_for(16, 0, "cfg+views+frames").frames == 4
