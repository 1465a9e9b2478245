"""pnc_concat_add_stats against pnc_concat_add + pnc_groupnorm_stats at the 12 concat sites of the UNet's output path (config 3)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import hip  # noqa: E402
from tools.exp.stagger_kbench import timed  # noqa: E402

DEV = "cuda"
F = 16
tot = [0.0, 0.0]
for (H, W, C1, C2) in [(4, 48, 1280, 1280)] * 3 + [(8, 96, 1280, 1280), (8, 96, 1280, 1280), (8, 96, 1280, 640), (16, 192, 1280, 640), (16, 192, 640, 640), (16, 192, 640, 320),
                       (32, 384, 640, 320), (32, 384, 320, 320), (32, 384, 320, 320)]:
    Npix, M, C = H * W, F * H * W, C1 + C2
    a, s_, c = torch.randn(M, C1, device=DEV), torch.randn(M, C2, device=DEV), torch.randn(M, C2, device=DEV)
    o32, o16 = torch.empty(M, C, device=DEV), torch.empty(M, C, device=DEV, dtype=torch.float16)
    lo = torch.empty(M, C, device=DEV, dtype=torch.uint8)
    ppc = max(16, min(128, Npix // 48), -(-Npix // 256))
    part1 = torch.empty(F * (-(-Npix // ppc)) * 96, device=DEV)
    part2 = torch.empty(F * (-(-Npix // ppc)) * 96, device=DEV)

    def two(i):
        hip.concat_add(a, C1, s_, c, C2, M, o32, o16, lo)
        hip.groupnorm_stats(o32, C, F, Npix, C, ppc, part1)

    def one(i):
        hip.concat_add(a, C1, s_, c, C2, M, o32, o16, lo, gn_part=part2, frames=F, ppc=ppc)
    t = timed({"two": two, "one": one}, iters=10, rounds=5)
    tot[0] += t["two"]; tot[1] += t["one"]
    print(f"{H}x{W} C1={C1} C2={C2}: concat + stats launch {t['two']:7.1f} us   concat with records {t['one']:7.1f} us", flush=True)
print(f"12 sites per step: {tot[0] / 1e3:.2f} ms -> {tot[1] / 1e3:.2f} ms")
