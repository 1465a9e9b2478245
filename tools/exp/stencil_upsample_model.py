"""Index model of the stencil-tile kernel for the nearest-x2 convs (Upsample: interpolate(x, 2) then conv3x3), DESIGN.md
section 12 item 4 — NOT built yet.  Checks, in plain Python, the arithmetic the kernel would use:
  * a 16x16 (8x32) OUTPUT tile at (Y0, X0) of the upsampled image reads the (TH/2 + 2) x (TW/2 + 2) INPUT pixels
    from (Y0/2 - 1, X0/2 - 1): halo row / column of output pixel (py, px), tap (ky, kx) = ((py+ky-1)>>1) + 1, ((px+kx-1)>>1) + 1;
  * padding of the UPSAMPLED image (u = -1, u = Hout) is the input rows -1 / Hin, i.e. the out-of-bounds halo rows;
  * LDS image: halo row r = hy * HW2 + hx, 128 B per row, 16-byte chunk XOR-swizzled by (hx >> 1) & 7: bank conflicts of the
    ds_read_b128 lane groups per tap (adjacent output pixels share an input pixel = the same address = a broadcast).
Run: python tools/exp/stencil_upsample_model.py"""
import itertools

import torch
import torch.nn.functional as TF

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]


def conv_by_tiles(x, w, TW):
    """x [Hin, Win, C] -> y [2 Hin, 2 Win, N] through the tile / halo index arithmetic"""
    Hin, Win, C = x.shape
    N = w.shape[0]
    Hout, Wout, TH = 2 * Hin, 2 * Win, 256 // TW
    HH, HW2 = TH // 2 + 2, TW // 2 + 2
    y = torch.zeros(Hout, Wout, N)
    for Y0, X0 in itertools.product(range(0, Hout, TH), range(0, Wout, TW)):
        halo = torch.zeros(HH, HW2, C)                       # out-of-image rows stay zero (PNC_BUF_OOB)
        for hy, hx in itertools.product(range(HH), range(HW2)):
            iy, ix = Y0 // 2 - 1 + hy, X0 // 2 - 1 + hx
            if 0 <= iy < Hin and 0 <= ix < Win:
                halo[hy, hx] = x[iy, ix]
        for py, px in itertools.product(range(TH), range(TW)):
            acc = torch.zeros(N)
            for ky, kx in itertools.product(range(3), range(3)):
                hy, hx = ((py + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1
                acc += w[:, :, ky, kx] @ halo[hy, hx]
            y[Y0 + py, X0 + px] = acc
    return y


def worst_conflict(TW):
    TH, HW2 = 256 // TW, TW // 2 + 2
    worst = 1
    for wm, i, ky, kx, c in itertools.product(range(4), range(2), range(3), range(3), range(8)):
        addr = {}
        for frow in range(32):
            R = wm * 64 + i * 32 + frow
            py, px = R // TW, R % TW
            hy, hx = ((py + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1
            addr[frow] = (hy * HW2 + hx) * 128 + ((c ^ ((hx >> 1) & 7)) << 4)
        for g in GROUPS:
            slots = {}
            for lane in g:
                slots.setdefault((addr[lane] // 16) % 16, set()).add(addr[lane])
            worst = max(worst, max(len(v) for v in slots.values()))
    return worst


if __name__ == "__main__":
    torch.manual_seed(0)
    for TW, Hin, Win in [(16, 8, 16), (32, 4, 32), (16, 16, 8)]:
        x, w = torch.randn(Hin, Win, 8), torch.randn(5, 8, 3, 3)
        ref = TF.conv2d(TF.interpolate(x.permute(2, 0, 1)[None], scale_factor=2, mode="nearest"), w, padding=1)[0].permute(1, 2, 0)
        got = conv_by_tiles(x, w, TW)
        print(f"TW={TW} image {Hin}x{Win}: max |tile model - conv2d(interpolate)| = {(got - ref).abs().max().item():.2e}; "
              f"halo {256 // TW // 2 + 2}x{TW // 2 + 2} = {(256 // TW // 2 + 2) * (TW // 2 + 2) * 128 / 1024:.1f} KB per slice; "
              f"worst ds_read_b128 conflict with the (hx>>1)&7 swizzle: {worst_conflict(TW)}-way")
