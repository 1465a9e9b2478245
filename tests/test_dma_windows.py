"""The GEMM kernels address their operands with 32-bit byte offsets from a per-tile window origin (buffer resources,
panacea_amd/csrc/gemm_kernel.h: a_window_origin / make_row / a_chunk_off; gemm_stencil_tile.hip: one frame per tile).
This restates that arithmetic in integers and checks, for every contraction shape of BASELINE config 3, the first-stage
encoder / decoder at 256x3072 and the ControlNet hint stem, that (1) window origin + offset is exactly the element the
gather means, for every tap at the tile corners, and (2) every offset stays below the out-of-bounds marker 2^31 — the
maximum sizes are where a 32-bit offset would wrap first."""
import itertools

import pytest

OOB = 0x80000000
F, T = 16, 8


def plain_offsets(M, lda, K, BM=256):
    for m0 in {0, (M // BM // 2) * BM, ((M - 1) // BM) * BM}:
        origin = m0 * lda
        for m in {m0, min(M - 1, m0 + BM - 1)}:
            for kc in {0, K - 8}:
                off = ((m - m0) * lda + kc) * 2
                assert off < OOB and origin + off // 2 == m * lda + kc
                yield off


def conv3x3_offsets(Fr, Hin, Win, Cin, stride=1, up=False, BM=256):
    Hout, Wout = (2 * Hin, 2 * Win) if up else ((Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1)
    hw, M = Hout * Wout, Fr * Hout * Wout
    frame = Hin * Win * Cin
    for m0 in {0, (M // BM // 2) * BM, ((M - 1) // BM) * BM}:
        f0 = m0 // hw
        origin = f0 * frame
        for m in {m0, min(M - 1, m0 + BM - 1)}:
            f, pix = divmod(m, hw)
            y, x = divmod(pix, Wout)
            rel = (f - f0) * frame
            for ky, kx, ci in itertools.product(range(3), range(3), {0, Cin - 8}):
                if up:
                    uy, ux = y + ky - 1, x + kx - 1
                    ok, iy, ix = (0 <= uy < Hout and 0 <= ux < Wout), uy >> 1, ux >> 1
                else:
                    iy, ix = y * stride + ky - 1, x * stride + kx - 1
                    ok = 0 <= iy < Hin and 0 <= ix < Win
                if not ok:
                    continue
                off = (rel + (iy * Win + ix) * Cin + ci) * 2
                assert 0 <= off < OOB and origin + off // 2 == f * frame + (iy * Win + ix) * Cin + ci
                yield off
    # the stencil-tile kernel: offsets inside ONE frame
    yield (frame - 8) * 2


def conv1d_offsets(B, Npix, C, BM=256):
    M = B * T * Npix
    for m0 in {0, (M // BM // 2) * BM, ((M - 1) // BM) * BM}:
        base_row = max(0, m0 - Npix)
        for m in {m0, min(M - 1, m0 + BM - 1)}:
            t = (m // Npix) % T
            for tap, ci in itertools.product(range(3), {0, C - 8}):
                if not 0 <= t + tap - 1 < T:
                    continue
                off = ((m - base_row) * C + (tap - 1) * Npix * C + ci) * 2
                assert 0 <= off < OOB and base_row * C + off // 2 == (m + (tap - 1) * Npix) * C + ci
                yield off


LEVELS = [(320, 32, 384), (640, 16, 192), (1280, 8, 96), (1280, 4, 48)]


def test_unet_and_controlnet_shapes_stay_far_below_the_marker():
    worst = 0
    for C, H, W in LEVELS:
        M = F * H * W
        for K in (C, 2 * C, 3 * C, 4 * C, 1024):                       # proj / skip / qkv source / ff2 / text keys
            worst = max(worst, max(plain_offsets(M, K, K)))
        for cin in (C, 2 * C, 3 * C):                                  # ResBlock convs incl. the decoder's concatenated inputs
            worst = max(worst, max(conv3x3_offsets(F, H, W, cin)))
        worst = max(worst, max(conv3x3_offsets(F, H, W, C, stride=2)), max(conv3x3_offsets(F, H, W, C, up=True)))
        worst = max(worst, max(conv1d_offsets(2, H * W, C)))
    # ControlNet hint stem: 19 (padded 24) -> 16 -> 16 -> 32 s2 -> 32 -> 96 s2 -> 96 -> 256 s2 -> 320 on 256x3072 frames
    for (h, w, cin, s) in [(256, 3072, 24, 1), (256, 3072, 16, 1), (256, 3072, 16, 2), (128, 1536, 32, 1), (128, 1536, 32, 2),
                           (64, 768, 96, 1), (64, 768, 96, 2), (32, 384, 256, 1)]:
        worst = max(worst, max(conv3x3_offsets(F, h, w, cin, stride=s)))
    assert worst < OOB // 8, worst                                     # > 8x head room on the denoising path


@pytest.mark.parametrize("frames", [8, 16])
def test_first_stage_shapes_at_full_resolution_fit(frames):
    """256x3072 frames x 256 channels are 403 MB each: the last pixels of such a frame are the largest offsets of the library."""
    worst = 0
    for (h, w, c) in [(256, 3072, 128), (256, 3072, 256), (128, 1536, 256), (128, 1536, 512), (64, 768, 512), (32, 384, 512)]:
        worst = max(worst, max(conv3x3_offsets(frames, h, w, c)))
        if h < 256:
            worst = max(worst, max(conv3x3_offsets(frames, h, w, c, up=True)))
        worst = max(worst, max(conv3x3_offsets(frames, h, w, c, stride=2)))
    # mid-block attention of one frame: scores [12288 x 12288] fp16 as the A operand of P V
    worst = max(worst, max(plain_offsets(12288, 12288, 12288)))
    assert worst < OOB, worst
    assert worst > OOB // 8          # ... and this IS the case that decides the marker: keep it covered
