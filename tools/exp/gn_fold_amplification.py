"""What folding the STT GroupNorm into proj_in would do to the operand (a measurement instead of a build).

proj_in(GN(x)) = (W diag(s_f)) x + (W t_f + b) with s = gamma * rstd, t = beta - mean * s per (frame, channel): the GEMM would consume the
RAW stream x (fp16 hi + lo) against per-frame scaled weights instead of the normalised y = s x + t.  Both operands are rounded
relative to their own magnitude, so the rounding error of the product grows by |s x| / |y - t| = sqrt(1 + (mean / sigma)^2) per
(frame, group).  This tool runs ONE evaluation of the full network at BASELINE config 3's shape on the GPU and reports that factor
at every GroupNorm-without-SiLU site (the 69 STT branch entries).  Usage: python tools/exp/gn_fold_amplification.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from panacea_amd import build_network, configs, engine as E, synth          # noqa: E402


def main():
    kw = configs.get("full")
    w = build_network(kw)
    manifest = json.loads((ROOT / "tests" / "golden" / "manifest_full.json").read_text())
    w.diffusion_model.load_state_dict(synth.synth_state_dict(manifest, salt=0), strict=True)
    w = w.to("cuda")
    inp = {k: v.to("cuda") for k, v in synth.synth_inputs(2, 8, 32, 384, context_dim=kw["context_dim"], t_index=500).items()}
    sites = []
    inner = E.gn_spatial

    def probe(rt, x32, F, N, C, gamma, beta, eps, silu, split=None, tail_rows=0):
        if not silu:
            X = x32.view(F, N, 32, C // 32).float()
            mean = X.mean(dim=(1, 3))
            var = X.var(dim=(1, 3), unbiased=False)
            sites.append((C, N, (mean.abs() / torch.sqrt(var + eps)).flatten()))
        return inner(rt, x32, F, N, C, gamma, beta, eps, silu, split=split, tail_rows=tail_rows)
    E.gn_spatial = probe
    import panacea_amd.nn.attention as A
    A.E.gn_spatial = probe
    with torch.no_grad():
        w(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "cond_feat")})
    torch.cuda.synchronize()
    print(f"{len(sites)} GroupNorm-without-SiLU sites (STT branch entries), |mean| / sigma per (frame, group):")
    allr = torch.cat([r for _, _, r in sites])
    amp = torch.sqrt(1 + allr * allr)
    q = torch.tensor([0.5, 0.9, 0.99], device=allr.device)
    print(f"  all sites: median {allr.median().item():.2f}  90 % {torch.quantile(allr, q[1]).item():.2f}  99 % {torch.quantile(allr, q[2]).item():.2f}"
          f"  max {allr.max().item():.2f}")
    print(f"  rounding-error growth sqrt(1 + r^2): rms {amp.pow(2).mean().sqrt().item():.2f}  mean {amp.mean().item():.2f}  max {amp.max().item():.2f}")
    by = {}
    for C, N, r in sites:
        by.setdefault((C, N), []).append(r)
    for (C, N), rs in sorted(by.items()):
        r = torch.cat(rs)
        a = torch.sqrt(1 + r * r)
        print(f"  C = {C:5d} N = {N:6d}: {len(rs):3d} sites, |mean|/sigma median {r.median().item():.2f} max {r.max().item():.2f}; growth rms {a.pow(2).mean().sqrt().item():.2f}")


if __name__ == "__main__":
    main()
