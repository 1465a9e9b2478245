"""One eps evaluation of the full-width network (B = 2 CFG halves, T = 8, 32x384 latent) unsharded and through the loop-back view
group (G = 1: every view-band code path — conv column exchange, combined GroupNorm records, neighbour-view attention layout — with
the exchanges served locally): the machinery's own cost on one device.  Usage: python tools/exp/view_loopback_time.py [iters] [both|unsharded|loopback]"""
import sys
import time
from pathlib import Path

import torch

import json

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, configs, engine as E, parallel, synth          # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    kw = configs.get("full")
    w = build_network(kw)
    manifest = json.loads((ROOT / "tests" / "golden" / "manifest_full.json").read_text())      # names and shapes of the state dict
    w.diffusion_model.load_state_dict(synth.synth_state_dict(manifest, salt=0), strict=True)
    w = w.to("cuda")
    inp = {k: v.to("cuda") for k, v in synth.synth_inputs(2, 8, 32, 384, context_dim=kw["context_dim"], t_index=500).items()}

    def cond(i):
        return {k: i[k] for k in ("concat", "crossattn", "cond_feat")}

    def timed(tag):
        with torch.no_grad():
            for _ in range(2):
                out = w(inp["x"], inp["t"], cond(inp))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                out = w(inp["x"], inp["t"], cond(inp))
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / iters
        print(f"{tag:28s} {ms:8.2f} ms / evaluation")
        return out, ms
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    vs = E.ViewShard(1, 0, None)
    if which != "both":             # one form only (for a kernel-stats profile of it)
        if which == "loopback":
            parallel.apply_view_shard(w, vs)
        timed(which)
        return
    ref, t_two = timed("unsharded, two streams")
    # sharded runs issue their collectives from ONE stream (controlmodel.py: same order on every rank): the fair base line
    for m in w.modules():
        if hasattr(m, "two_stream"):
            m.two_stream = False
    ref, t_ref = timed("unsharded, one stream")
    parallel.apply_view_shard(w, vs)
    got, t_vs = timed("view loop-back (G = 1)")
    d = (got - ref).abs()
    print(f"view machinery: +{t_vs - t_ref:.2f} ms ({100 * (t_vs / t_ref - 1):.1f} %), {vs.exchanges // (iters + 2)} exchanges / evaluation, "
          f"eps difference max {d.max().item():.3e} mean {d.mean().item():.3e}")
    # round 5: the ControlNet on its side stream over a shard object (= process group) of its own
    for m in w.modules():
        if hasattr(m, "two_stream"):
            m.two_stream = True
    vs2 = E.ViewShard(1, 0, None)
    parallel.apply_view_shard(w, vs, vs2)
    got2, t_vs2 = timed("view loop-back, two streams")
    print(f"view machinery on two streams: {t_vs2:.2f} ms = +{t_vs2 - t_two:.2f} ms over the unsharded two-stream evaluation; eps bit-identical to the one-stream "
          f"loop-back: {bool(torch.equal(got2, got))}")


if __name__ == "__main__":
    main()
