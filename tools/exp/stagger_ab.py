"""Whole-network A/B of PNC_OPT_GEMM_STAGGER in ONE process (round 5): the full-size network is built once, then for each option value
(0 = round 4's loops, 8 = default threshold, 1 = always) (a) network evaluations are timed in interleaved rounds on two streams, as the
bench does, and (b) one evaluation per value is profiled per GEMM shape (HIP events around every launch, one stream); the shape table
prints the values side by side.

    python tools/exp/stagger_ab.py [values, default 0,8,1] [rounds, default 3]
"""
import collections
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, configs, hip, synth  # noqa: E402

vals = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8,1").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
opt = getattr(hip, sys.argv[3]) if len(sys.argv) > 3 else hip.OPT_GEMM_STAGGER
kw = configs.get("full")
man = json.loads((ROOT / "tests/golden/manifest_full.json").read_text())
net = build_network(kw)
net.diffusion_model.load_state_dict(synth.synth_state_dict(man), strict=True)
net = net.to("cuda")
B, T, h, w = configs.SHAPES["full"]
g = {k: v.to("cuda") for k, v in synth.synth_inputs(B, T, h, w, context_dim=kw["context_dim"]).items()}
c = {k: g[k] for k in ("concat", "crossattn", "cond_feat")}


def evals(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net(g["x"], g["t"], c)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = None
    for v in vals:
        hip.set_option(opt, v)
        e = net(g["x"], g["t"], c)
        if ref is None:
            ref = e
        print(f"value {v}: eps bit-identical to value {vals[0]}: {torch.equal(e, ref)}", flush=True)
    evals(3)
    times = collections.defaultdict(list)
    for r in range(rounds):
        for v in vals:
            hip.set_option(opt, v)
            times[v].append(evals(8))
    for v in vals:
        ts = sorted(times[v])
        print(f"value {v}: network evaluation (two streams) min {ts[0]:.2f} median {ts[len(ts) // 2]:.2f} ms  {['%.2f' % t for t in times[v]]}", flush=True)
    # per-shape profile
    net.diffusion_model.two_stream = False
    orig = hip.gemm
    prof = {}
    for v in vals:
        hip.set_option(opt, v)
        rec = []

        def gemm(a16, w16, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); orig(a16, w16, **k); e1.record()
            tag = ("geglu " if k.get("geglu") else "") + ("res " if k.get("res1") is not None else "") + ("T " if k.get("out16t") is not None else "") + \
                ("o32 " if k.get("out32") is not None else "") + ("o16" if k.get("out16") is not None else "") + \
                ("" if k.get("a16_lo") is None else (" lo8" if k["a16_lo"].dtype == torch.uint8 else " lo16")) + (" ln" if k.get("ln_out16") is not None else "")
            rec.append(((k.get("a_mode", 0), k["M"], k["N"], k["K"], tag), e0, e1))
        hip.gemm = gemm
        net(g["x"], g["t"], c)
        hip.gemm = orig
        torch.cuda.synchronize()
        agg = collections.defaultdict(lambda: [0, 0.0])
        for key, e0, e1 in rec:
            agg[key][0] += 1
            agg[key][1] += e0.elapsed_time(e1)
        prof[v] = agg
    base = prof[vals[0]]
    print("gemm-family total per evaluation (one stream): " + "  ".join(f"[{v}] {sum(x[1] for x in prof[v].values()):.1f} ms" for v in vals))
    for key, (n, ms) in sorted(base.items(), key=lambda kv: -kv[1][1])[:48]:
        mode, M, N, K, tag = key
        line = f"mode{mode} M={M:7d} N={N:5d} K={K:5d} {tag:24s} x{n:3d}"
        for v in vals:
            m2 = prof[v][key][1]
            line += f"  [{v}] {m2 / n * 1e3:7.1f} us {2.0 * M * N * K * n / m2 / 1e9:6.0f} TF"
        print(line)
