"""pnc_ff_chain_f16 — x = ff(norm3(x)) + x of a BasicTransformerBlock in one launch (attention.py:91-117, 726-747) — on the
MI355X against the torch emulation (which unpacks the weight tape again: the packing is under test too) and against the
unfused launch sequence of the same library (LayerNorm + GEGLU GEMM + output GEMM)."""
import pytest
import torch

import emu
from panacea_amd import engine, hip
from test_kernels_gpu import DEV, check, rnd

pytestmark = pytest.mark.gpu
C = 320


def _problem(M, inner, seed=0):
    w1 = rnd(2 * inner, C, scale=C ** -0.5, seed=seed + 1).half().float()
    w2 = rnd(C, inner, scale=0.5 * inner ** -0.5, seed=seed + 2).half().float()
    return dict(x=rnd(M, C, seed=seed + 3) * 1.5 + 0.2, w1=w1, w2=w2, b1=rnd(2 * inner, seed=seed + 4) * 0.1, b2=rnd(C, seed=seed + 5) * 0.1,
                g=rnd(C, seed=seed + 6) * 0.2 + 1.0, b=rnd(C, seed=seed + 7) * 0.1, tape=engine.pk_ff_chain(w1, w2))


@pytest.mark.parametrize("M,inner", [(128, 1280), (1024, 1280), (384, 256), (256, 1536)])
def test_ff_chain_matches_the_emulation(M, inner):
    q = _problem(M, inner)
    outs = {}
    for name, be in (("hip", hip), ("emu", emu)):
        o32 = torch.zeros(M, C, device=DEV)
        o16 = torch.zeros(M, C, device=DEV, dtype=torch.float16)
        lo8 = torch.zeros(M, C, device=DEV, dtype=torch.uint8)
        be.ff_chain(q["x"], C, M, C, inner, q["g"], q["b"], 1e-5, q["tape"], q["b1"], q["b2"], out32=o32, ldo32=C, out16=o16, ldo16=C,
                    out16_lo=lo8)
        outs[name] = (o32, o16, lo8)
    torch.cuda.synchronize()
    h, e = outs["hip"], outs["emu"]
    # The fp16 rounding of a hidden unit falls on the other side of a tie in ~0.4 % of the cases (tabulated Phi and another
    # summation order move the fp32 value by ~1e-6 relative): ~5 of a row's 1280 hidden units, each worth one fp16 ulp of h times a
    # weight ~ 3e-5 on an output.  So: the MEAN difference is at fp32-noise level, the maximum a few flips wide.
    d = (h[0] - e[0]).abs()
    print(f"ff_chain vs emulation: max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.mean().item() <= 4e-5
    check("out32", h[0], e[0], 8e-4, 0)
    check("out16", h[1], e[1], 4e-3)
    rec = h[1].float() + h[2].view(torch.float8_e4m3fn).float() / 2048.0
    assert ((rec - h[0]).abs() <= 2.0 ** -14 * h[0].abs() + 2.0 ** -20).all()          # the pair carries the kernel's own fp32 value
    assert (h[0] - q["x"]).abs().max() > 0.05                                           # (the feed-forward did contribute)


def test_ff_chain_in_place_deterministic_and_vs_unfused_launches():
    M, inner = 2048, 1280
    q = _problem(M, inner, seed=10)
    # in place on the stream + fp16 lo plane, twice: same bits
    runs = []
    for _ in range(2):
        x = q["x"].clone()
        o16 = torch.zeros(M, C, device=DEV, dtype=torch.float16)
        lo = torch.zeros(M, C, device=DEV, dtype=torch.float16)
        hip.ff_chain(x, C, M, C, inner, q["g"], q["b"], 1e-5, q["tape"], q["b1"], q["b2"], out32=x, ldo32=C, out16=o16, ldo16=C, out16_lo=lo)
        runs.append((x, o16, lo))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    # the unfused sequence of the same library: LayerNorm kernel -> GEGLU GEMM -> output GEMM (+ residual)
    w1i, b1i = engine.pk_geglu(q["w1"], q["b1"])
    a16 = torch.zeros(M, C, device=DEV, dtype=torch.float16)
    hid = torch.zeros(M, inner, device=DEV, dtype=torch.float16)
    ref = q["x"].clone()
    hip.layernorm(q["x"], C, M, C, q["g"], q["b"], 1e-5, a16, C)
    hip.gemm(a16, w1i, M=M, N=2 * inner, K=C, lda=C, bias=b1i, geglu=True, out16=hid, ldc16=inner)
    hip.gemm(hid, q["w2"].half(), M=M, N=C, K=inner, lda=inner, bias=q["b2"], res1=ref, ldr1=C, out32=ref, ldc32=C)
    torch.cuda.synchronize()
    d = (runs[0][0] - ref).abs()
    print(f"fused vs unfused launches: max {d.max().item():.3e} mean {d.mean().item():.3e}")
    assert d.max().item() <= 8e-4 and d.mean().item() <= 4e-5


def test_ff_chain_argument_checks():
    q = _problem(128, 1280)
    o = torch.zeros(128, C, device=DEV)
    assert hip.ff_chain_supported(128, 320, 1280) and not hip.ff_chain_supported(100, 320, 1280)
    assert not hip.ff_chain_supported(128, 640, 2560) and not hip.ff_chain_supported(128, 320, 1312)
    with pytest.raises(hip.PncError):            # rows not a multiple of 128
        hip.ff_chain(q["x"], C, 100, C, 1280, q["g"], q["b"], 1e-5, q["tape"], q["b1"], q["b2"], out32=o, ldo32=C)
    with pytest.raises(hip.PncError):            # a tape of another hidden width
        hip.ff_chain(q["x"], C, 128, C, 1024, q["g"], q["b"], 1e-5, q["tape"], q["b1"], q["b2"], out32=o, ldo32=C)
    with pytest.raises(hip.PncError):            # no output
        hip.ff_chain(q["x"], C, 128, C, 1280, q["g"], q["b"], 1e-5, q["tape"], q["b1"], q["b2"])


def test_network_with_fused_feed_forward_meets_the_contract():
    """The full-width network with its level-0 feed-forwards on the fused launch (engine.FUSE_FF_CHAIN, off by default) against the
    oracle: same 1e-3 contract as the default path, and it does change the launch sequence (different bits)."""
    from helpers import cond, err_stats, oracle_cfg, product_network, step_inputs
    from oracle import panacea_oracle as po
    from panacea_amd import configs
    kw = configs.with_frames(configs.get("full"), 2)
    w, sd, _ = product_network("full", "cpu", kw=kw)
    inp = step_inputs("full", kw, "cpu", shape=(1, 2, 16, 192))
    ref = po.wrapper_forward(sd, oracle_cfg(kw), inp["x"], inp["t"], cond(inp))
    w = w.to(DEV)
    g = {k: v.to(DEV) for k, v in inp.items()}
    base = w(g["x"], g["t"], cond(g))
    assert engine.FUSE_FF_CHAIN is False
    engine.FUSE_FF_CHAIN = True
    try:
        eps = w(g["x"], g["t"], cond(g))
    finally:
        engine.FUSE_FF_CHAIN = False
    st = err_stats(eps, ref)
    print("full network, 16x192, fused feed-forward:", st)
    assert not torch.equal(eps, base)
    assert st["max_abs"] <= 1e-3 and st["mean_abs"] <= 2e-4, st
