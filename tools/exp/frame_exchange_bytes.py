"""Bytes a rank of a frame group sends per denoising step at BASELINE config 3 (one CFG half per rank: B = 1, T = 8, 32x384 latent),
from the module tree of the Panacea+ network — the accounting of engine.FrameShard, evaluated without running the network:

  ResBlock3D temporal site, round 2 form ("transpose"): fp32 h to the pixel sharding and back = 2 x T_l (G-1)/G x N C 4 B
  ResBlock3D temporal site, round 4 form ("halo"):      all-reduce of the [N, 32, 2] fp32 partial sums (ring: 2 (G-1)/G x payload)
                                                        + one frame of the fp16 operand and its e4m3 lo plane per neighbour (3 B)
  STT temporal branch (both rounds):                    GroupNorm output in, last block's output back: fp16 + e4m3 lo, transposed

    python tools/exp/frame_exchange_bytes.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from panacea_amd import build_network, configs  # noqa: E402
from panacea_amd.nn.attention import SpatialTemporalTransformer  # noqa: E402
from panacea_amd.nn.openaimodel import Downsample, ResBlock3D, Upsample  # noqa: E402


def walk(net, T=8, h=32, w=384):
    """-> [(kind, pixels per frame, channels)] in execution order for the UNet and the ControlNet"""
    out = []
    for root in (net, net.controlnet):
        n = h * w
        seqs = list(root.input_blocks) + [root.middle_block] + (list(root.output_blocks) if hasattr(root, "output_blocks") else [])
        for seq in seqs:
            for m in seq:
                if isinstance(m, ResBlock3D):
                    out += [("res", n, m.out_channels)] * 2
                elif isinstance(m, SpatialTemporalTransformer):
                    out.append(("stt", n, m.in_channels))
                elif isinstance(m, Downsample):
                    n //= 4
                elif isinstance(m, Upsample):
                    n *= 4
    return out


def main():
    net = build_network(configs.get("full")).diffusion_model
    sites = walk(net)
    T = 8
    print(f"{sum(k == 'res' for k, _, _ in sites)} ResBlock3D temporal sites, {sum(k == 'stt' for k, _, _ in sites)} STT temporal branches")
    for G in (2, 4):
        tl = T // G
        frac = (G - 1) / G
        nb_avg = 2 * (G - 1) / G                      # neighbours per rank, averaged over the ranks of the group
        tr = sum(2 * tl * frac * n * c * 4 for k, n, c in sites if k == "res")
        halo = sum(nb_avg * n * c * 3 + 2 * frac * n * 64 * 4 for k, n, c in sites if k == "res")
        stt = sum(2 * tl * frac * n * c * 3 for k, n, c in sites if k == "stt")
        print(f"G = {G}: ResBlock sites transpose {tr / 1e9:.3f} GB | halo {halo / 1e9:.3f} GB;  STT temporal {stt / 1e9:.3f} GB;  "
              f"per rank and step: round 2 {(tr + stt) / 1e9:.3f} GB -> round 4 {(halo + stt) / 1e9:.3f} GB")


if __name__ == "__main__":
    main()
