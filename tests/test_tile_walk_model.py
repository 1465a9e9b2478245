"""Index model of the persistent FF1 kernel's tile walk (panacea_amd/csrc/gemm_kernel.h: gemm_geglu_persist_kernel::tile_origin +
common.h: xcd_remap): every output tile is visited exactly once, the workgroups of one XCD (blockIdx % 8) stay inside one contiguous
range of tile ids, and consecutive tile ids of a range walk `group_m` row panels before moving to the next column tile — the properties the
kernel's L2 reuse argument rests on.  (The device code itself is checked bit for bit against the one-tile-per-workgroup kernel by
tests/test_kernels_gpu.py::test_gemm_geglu_persistent_kernel_is_bit_identical.)"""
import pytest


def xcd_remap(b, nblk):
    q, r = nblk >> 3, nblk & 7
    xcd, idx = b & 7, b >> 3
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + idx


def tile_origin(v, tiles_m, tiles_n, group_m):
    tile = xcd_remap(v, tiles_m * tiles_n)
    if group_m > 0:
        width = group_m * tiles_n
        gid = tile // width
        first_m = gid * group_m
        gsz = min(tiles_m - first_m, group_m)
        inn = tile - gid * width
        return first_m + inn % gsz, inn // gsz
    return tile // tiles_n, tile % tiles_n


@pytest.mark.parametrize("M,N,grid", [(196608, 2560, 256), (49152, 5120, 256), (12288, 10240, 256), (256 * 70, 2560, 256),
                                      (256 * 26, 5120, 240), (256 * 64, 2560, 304)])
def test_persistent_walk_visits_every_tile_once(M, N, grid):
    tiles_m, tiles_n = M // 256, N // 256
    ntile = tiles_m * tiles_n
    group_m = min(4, tiles_m) if tiles_n > 8 else 0
    seen = {}
    for b in range(min(grid, ntile)):
        v = b
        while v < ntile:
            t = tile_origin(v, tiles_m, tiles_n, group_m)
            assert t not in seen, (t, b, seen[t])
            assert 0 <= t[0] < tiles_m and 0 <= t[1] < tiles_n
            seen[t] = b
            v += grid
    assert len(seen) == ntile
    if grid % 8 == 0:
        # a workgroup never leaves its XCD's contiguous range of tile ids
        q, r = ntile >> 3, ntile & 7
        for b in range(min(grid, ntile)):
            x = b & 7
            lo = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
            hi = lo + (q + 1 if x < r else q)
            v = b
            while v < ntile:
                assert lo <= xcd_remap(v, ntile) < hi
                v += grid
