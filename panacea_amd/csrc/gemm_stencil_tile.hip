// gemm_stencil_tile.hip — 3x3 conv (stride 1, pad 1, Cin % 64 == 0) as implicit GEMM over SPATIAL output tiles that
// contain their stencil neighbours.
//
// The per-tap gather (gemm_conv3x3.hip) DMAs, for every 64-channel slice of the input, nine A tiles — one per tap — that are
// the same pixels shifted by one: 9 x 32 KB per 256 output pixels, every byte of it through the L2 -> LDS path whose cost
// adds to the MFMAs' (DESIGN.md §4 "where the time goes").  Here a workgroup owns a TH x TW block of output pixels of one
// frame (16 x 16, or 8 x 32 where the image height is no multiple of 16), stages the block plus its halo —
// (TH + 2) x (TW + 2) pixels, 41-43 KB — ONCE per slice, and the nine taps read their A fragments from it at shifted LDS
// rows.  Per slice and 256 x 320 output tile: 41 + 9 x 40 KB instead of 9 x 32 + 9 x 40 KB — 1.6x fewer DMA bytes, same MFMAs,
// same LDS fragment reads.  Measured (MI355X, profiles/round2/kbench_r2k_stencil_tiles.log): level-0 convs 790-925 ->
// 915-1050 TFLOP/s, first-stage decoder convs 800 -> 960-1020.
// (The temporal k = 3 conv was built on the same scheme — 32 pixels x all 8 frames per tile, 1.4x fewer DMA bytes — and
// measured no faster: at K = 3 C its time is the two-stream fp32 epilogue's, not the operand path's.  Not kept.)
//
// BN = NI x 64 (320 where N is a multiple of 320, else 256), 8 waves as 4 x 2, wave tile 64 x (NI x 32).
// LDS: two halo buffers (slice c + 1 arrives, one 1-KB piece per wave and iteration, while slice c is consumed) and a ring of
// three W HALF tiles — 32 of the 64 channels of one (slice, tap) pair, BN rows x 64 B: 2 x 41 + 3 x 20 = 142 KB at BN = 320
// (whole 64-channel W stages would need 2 x 41.5 + 2 x 40 = 163 KB).  Counted vmcnt waits + one raw s_barrier per half
// tile, placed in the middle of its MFMA stream (software pipeline below): two half tiles in flight, one landed.
// The K order (slice, tap, channel) is the per-tap kernel's, so are the products: results are bit-identical to its
// (precise operands included: the lo plane's pass runs first, the accumulators are scaled by 2^-11, then the hi plane's).
// The epilogue is gemm_kernel.h's epi_fast with a row map (RowHalo): tile-local row -> pixel of the frame.
#include "gemm_kernel.h"

namespace pnc_gemm {

template <int TWS, int NI, unsigned EPI>     // TW = 2^TWS columns per spatial tile; BN = NI * 64
__global__ __launch_bounds__(512) void stencil_tile_kernel(const PncGemmParams pin, const int group_m, const int nfull, const int tail_f,
                                                           const int stagger) {
    const PncGemmParams& p = pin;
    constexpr int TW = 1 << TWS, TH = 256 / TW;
    constexpr int IPS = 18;                                 // half-tile iterations per 64-channel slice: 9 taps x 2
    constexpr int HW2 = TW + 2, HROWS = (TH + 2) * HW2;
    constexpr int HBLK = (HROWS + 7) / 8;                   // 1-KB DMA pieces (8 halo rows of 128 B)
    constexpr int HBYTES = HBLK * 1024;
    constexpr int BN = NI * 64, NW = 8, WGN = 2, MI = 2;
    constexpr int WHB = BN * 64;                            // bytes of a W half tile: BN rows x 32 channels
    constexpr int WBLK = WHB / 1024;                        // its 1-KB DMA pieces (16 W rows x 64 B each): 16 / 20
    constexpr int W_IT = (WBLK + NW - 1) / NW;
    constexpr int H_IT = (HBLK + NW - 1) / NW;              // halo pieces per wave and slice: 6
    constexpr int ENI = 2, EPITCH = ENI * 32 + 4;
    static_assert(HW2 % 2 == 0, "the halo swizzle takes the address parity from the halo column");
    static_assert(2 * HBYTES + 3 * WHB <= 160 * 1024, "LDS budget");
    static_assert(2 * HBYTES + 3 * WHB >= NW * 32 * EPITCH * 4, "epilogue staging fits the operand buffers");
    static_assert(H_IT <= IPS, "one halo piece per wave and half-tile iteration");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wring = smem + 2 * HBYTES;

    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ A_lo = reinterpret_cast<const half_t*>(p.A_lo);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);

    const int tiles_x = p.Wout >> TWS, per_frame = (p.Hout / TH) * tiles_x;
    const int tiles_m = (p.M / (p.Hout * p.Wout)) * per_frame, tiles_n = (p.N + BN - 1) / BN;
    // Tail split (as in gemm_kernel.h): the tiles of a sparse last round are each run by tail_f workgroups that own 256 / tail_f of
    // the tile's pixels (whole tile rows); the waves of the other rows skip their reads, MFMAs and epilogue, only the halo rows
    // the part needs are staged, all waves still stage W.  Rows of a GEMM are independent: bit-identical to the unsplit launch.
    int tile, part = 0;
    if ((int)blockIdx.x < nfull) {
        tile = xcd_remap(blockIdx.x, nfull);
    } else {
        const int j = (int)blockIdx.x - nfull;
        tile = nfull + j / tail_f; part = j - (j / tail_f) * tail_f;
    }
    const bool split = (int)blockIdx.x >= nfull && tail_f > 1;
    int tn, tm;
    if (group_m > 0) {
        const int width = group_m * tiles_n;
        const int gid = tile / width, first_m = gid * group_m;
        const int gsz = min(tiles_m - first_m, group_m);
        const int in = tile - gid * width;
        tm = first_m + in % gsz; tn = in / gsz;
    } else {
        tn = tile % tiles_n; tm = tile / tiles_n;
    }
    const int f = tm / per_frame, trem = tm - f * per_frame;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int Y0 = ty * TH, X0 = tx << TWS, n0 = tn * BN;       // halo row hy / column hx = image row Y0 - 1 + hy, column X0 - 1 + hx
    const int64_t img_base = (int64_t)f * p.Hin * p.Win * p.Cin;
    const int base_m = (f * p.Hout + Y0) * p.Wout + X0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int rows_lo = split ? part * (256 / tail_f) : 0, rows_hi = split ? rows_lo + 256 / tail_f : 256;
    const bool wave_on = (wm * 64 >= rows_lo) && (wm * 64 < rows_hi);
    // halo pieces (8 halo rows each) that hold tile rows rows_lo / TW - 1 .. rows_hi / TW: the others are never read
    const int pc_lo = ((rows_lo >> TWS) * HW2) >> 3, pc_hi = (((rows_hi >> TWS) + 2) * HW2 + 7) >> 3;

    // Operands reach LDS through buffer resources (common.h: glds16_buf; 4-7 % over per-lane 64-bit pointers on every conv shape):
    // base = this tile's frame of each activation plane / the tile's first weight row, per-lane 32-bit byte offsets that do not
    // change along K (the slice / half tile enters as the scalar offset); PNC_BUF_OOB offsets read as zero (padding, N tail).
    // x_halo_off (a view band): image columns -1 and Win are read from the block [2][frames][Hin][Cin] at A + x_halo_off, beyond the frame
    const bool xh = p.x_halo_off != 0;
    const unsigned frame_bytes = xh ? 0x7FFFFF00u : (unsigned)(p.Hin * p.Win * p.Cin) * 2u;
    const int64_t xh_rel = p.x_halo_off - img_base + (int64_t)f * p.Hin * p.Cin;      // this frame's rows of the left column
    const int64_t xh_side = (int64_t)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Cin;        // ... of the right column, from there
    const buffer_rsrc_t rs_a = make_rsrc(A + img_base, frame_bytes);
    const buffer_rsrc_t rs_lo = make_rsrc((A_lo ? A_lo : A) + img_base, frame_bytes);
    const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);

    // ---- halo DMA: piece b = halo rows 8b .. 8b+7; lane l fills slot (l&7) of row 8b + (l>>3) with the source chunk
    // slot ^ ((hx>>1)&7), hx = the row's halo COLUMN.  The 16 lanes of a ds_read_b128 group read 16 consecutive pixels of
    // one or two tile rows = 16 consecutive halo columns (whatever the tap), i.e. all 16 (parity, hx>>1) pairs.
    auto halo_off = [&](int b) -> unsigned {            // this lane's byte offset of piece b inside the frame (the same for every slice)
        const int hr = b * 8 + (lane >> 3);
        const int hy = hr / HW2, hx = hr - hy * HW2;
        const int c8 = (lane & 7) ^ ((hx >> 1) & 7);
        const int iy = Y0 - 1 + hy, ix = X0 - 1 + hx;
        const bool yok = (hr < HROWS) && (iy >= 0) && (iy < p.Hin), xin = (ix >= 0) && (ix < p.Win);
        const bool ok = yok && (xin || (xh && ix >= -1 && ix <= p.Win));
        unsigned off = ok ? (unsigned)((iy * p.Win + ix) * p.Cin + c8 * 8) * 2u : PNC_BUF_OOB;           // out of the image: zeros
        if (ok && !xin) off = (unsigned)(xh_rel + (ix < 0 ? 0 : xh_side) + iy * p.Cin + c8 * 8) * 2u;
        return off;
    };
    auto request_halo = [&](bool lo_plane, int cc, int buf, int b, unsigned off) {
        glds16_buf(lo_plane ? rs_lo : rs_a, off, (unsigned)cc << 7, halo + buf * HBYTES + b * 1024);
    };
    auto issue_halo = [&](bool lo_plane, int cc, int buf, int b) { request_halo(lo_plane, cc, buf, b, halo_off(b)); };
    // ---- W DMA: half tile k = 32 channels of K tile k/2 = k offset 32 k of the packed [N][(ci/64, tap, ci%64)] weights.
    // LDS row R (128 B) = W rows 2R, 2R+1; slot = (n&1)*4 + (c ^ ((R>>1)&3)), c = 16-byte chunk of the 64-byte half row.
    const int nW = WBLK / NW + (wave < (WBLK % NW) ? 1 : 0);      // DMA instructions of this wave per half tile
    unsigned woff[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int R = (wave + NW * i) * 8 + (lane >> 3), slot = lane & 7;
        const int nl = 2 * R + (slot >> 2);
        const int c4 = (slot & 3) ^ ((R >> 1) & 3);
        woff[i] = (n0 + nl < p.N) ? (unsigned)(nl * p.ldw + c4 * 8) * 2u : PNC_BUF_OOB;
    }
    auto issue_w = [&](int k, int stage) {
        char* sb = wring + stage * WHB + wave * 1024;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            if (wave + NW * i < WBLK) glds16_buf(rs_w, woff[i], (unsigned)k << 6, sb + i * (NW * 1024));
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // this lane's fragment rows.  A: tile-local output row R -> halo row / column of tap 0; B: W row -> LDS row, slot
    const int frow = lane & 31, fk = lane >> 5;
    int hp0[MI], hx0[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int R = wm * 64 + i * 32 + frow;
        hx0[i] = R & (TW - 1);
        hp0[i] = (R >> TWS) * HW2 + hx0[i];
    }
    const int b_row = ((wn * (NI * 32) + frow) >> 1) * 128 + ((frow & 1) << 6);
    const int b_swz = (frow >> 2) & 3;                 // (R>>1)&3: blocks of 32 W rows shift R by 16

    // Fragments of one k-step (16 channels) of half tile (hbuf, tap, half, stage) -> register buffer b.  Explicit
    // ds_read_b128: the compiler's counter model waits lgkmcnt(0) across the loop's back edge, which would expose the latency of
    // the reads just issued (A/B on the device: 2-4 % at long K); the waits for these reads are written out in the loop below.
    half8v af[2][MI], bf[2][NI];
    // LDS addresses of those fragments (the pipeline computes them one batch AHEAD of the reads: round 6) ...
    auto frag_addr = [&](int hbuf, int tap, int half, int stage, int ks, unsigned (&a_addr)[MI], unsigned& b_addr) {
        const char* sa = halo + hbuf * HBYTES;
        const char* sb = wring + stage * WHB + b_row;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int toff = ky * HW2 + kx;
        const int c4 = ks * 2 + fk;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            a_addr[i] = (unsigned)(uintptr_t)(sa + (hp0[i] + toff) * 128 + (((half * 4 + c4) ^ (((hx0[i] + kx) >> 1) & 7)) << 4));
        b_addr = (unsigned)(uintptr_t)(sb + ((c4 ^ b_swz) << 4));
    };
    // ... and the reads
    auto frag_read = [&](const unsigned (&a_addr)[MI], unsigned b_addr, auto b_) {
        constexpr int b = decltype(b_)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[b][i]) : "v"(a_addr[i]));
#define PNC_STENCIL_RD_B(J)                                                                                            \
    if constexpr (NI > J)                                                                                              \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[b][J < NI ? J : 0]) : "v"(b_addr), "n"(J * 16 * 128));
        PNC_STENCIL_RD_B(0) PNC_STENCIL_RD_B(1) PNC_STENCIL_RD_B(2) PNC_STENCIL_RD_B(3) PNC_STENCIL_RD_B(4)
#undef PNC_STENCIL_RD_B
    };
    auto frags = [&](int hbuf, int tap, int half, int stage, int ks, auto b_) {
        unsigned a_addr[MI], b_addr;
        frag_addr(hbuf, tap, half, stage, ks, a_addr, b_addr);
        frag_read(a_addr, b_addr, b_);
    };
    auto mfmas = [&](auto b_) {
        constexpr int b = decltype(b_)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
    };
    const std::integral_constant<int, 0> B0{};
    const std::integral_constant<int, 1> B1{};

    // counted wait: everything but this wave's most recent W group (nW instructions) has landed
    auto wait_all_but_last_w = [&]() {
        if (nW == W_IT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_IT - 1) : "memory");
    };

    // Slices in execution order: with a precise operand the lo plane's nslices first, then the hi plane's
    const int nslices = p.Cin >> 6, nq1 = nslices * IPS;              // half tiles of one pass
    const int ns_tot = A_lo ? 2 * nslices : nslices, nq = ns_tot * IPS;
    auto slice_plane = [&](int gs) { return A_lo && gs < nslices; };      // true: the lo plane
    auto slice_cc = [&](int gs) { return gs >= nslices ? gs - nslices : gs; };

    // STAGGERED schedule (round 5, PNC_OPT_GEMM_STAGGER; gemm_kernel.h has the story): one PHASE per k-step —
    //     fragment reads of the k-step [+ DMA in the odd phases] | s_barrier | MI x NI MFMAs | s_barrier —
    // with waves 4-7 one barrier behind waves 0-3.  The odd phase of half tile q issues one halo piece of the next slice (as the
    // pipeline below does) and W half tile q + 2 into the ring stage of q - 1: every wave of BOTH groups finished its reads of that
    // stage two barriers earlier (its last reads sat in phase (q - 1, 1); the other group's lgkmcnt(0) after the first barrier of
    // that phase is passed by the time this group is behind the second barrier of phase (q, 0)).  The counted wait in the same
    // phase leaves only the W group just issued in flight: W(q + 1), read from the next phase on, has landed.  Same K order
    // and MFMA order per accumulator: bit-identical to the pipeline below.
    if (stagger == 1) {
        const int grp = wave >> 2;
#pragma unroll
        for (int i = 0; i < H_IT; ++i)
            if (wave + NW * i >= pc_lo && wave + NW * i < pc_hi && wave + NW * i < HBLK) issue_halo(slice_plane(0), 0, 0, wave + NW * i);
        issue_w(0, 0);
        if (nq > 1) issue_w(1 % nq1, 1);
        if (nq > 2) issue_w(2 % nq1, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();
        auto bar1 = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto bar2 = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        };
        int st = 0, gs = 0, r = 0, w2 = 3 % nq1;              // w2 = (q + 2) mod nq1 at the issue point of iteration q >= 1
        for (int q = 0; q < nq; ++q) {
            if (wave_on) frags(gs & 1, r >> 1, r & 1, st, 0, B0);
            bar1();
            if (wave_on) mfmas(B0);
            bar2();
            if (wave_on) frags(gs & 1, r >> 1, r & 1, st, 1, B1);
            if (r < H_IT && gs + 1 < ns_tot && wave + NW * r >= pc_lo && wave + NW * r < pc_hi && wave + NW * r < HBLK)
                issue_halo(slice_plane(gs + 1), slice_cc(gs + 1), (gs + 1) & 1, wave + NW * r);
            if (q >= 1) {
                if (q + 2 < nq) {
                    issue_w(w2, st == 0 ? 2 : st - 1);         // the stage of half tile q - 1 = (q + 2) mod 3
                    wait_all_but_last_w();
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                w2 = (w2 + 1 == nq1) ? 0 : w2 + 1;
            }
            bar1();
            if (wave_on) mfmas(B1);
            bar2();
            if (A_lo && q + 1 == nq1) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] *= LO_INV;
            }
            if (++r == IPS) { r = 0; ++gs; }
            st = (st == 2) ? 0 : st + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {
    // Software pipeline over half tiles q (two k-steps each), the barrier in the MIDDLE of q's MFMA stream:
    //   reads(q, ks1) | MFMA(q, ks0) | wait: W(q+1) landed, own reads of q done | s_barrier | DMA: halo piece, W(q+3) -> stage of q |
    //   reads(q+1, ks0) | MFMA(q, ks1)
    // so every fragment read runs under the other k-step's MFMAs and three half tiles are landed / in flight.
#pragma unroll
    for (int i = 0; i < H_IT; ++i)
        if (wave + NW * i >= pc_lo && wave + NW * i < pc_hi && wave + NW * i < HBLK) issue_halo(slice_plane(0), 0, 0, wave + NW * i);
    issue_w(0, 0);
    issue_w(1, 1);
    wait_all_but_last_w();
    __builtin_amdgcn_s_barrier();
    issue_w(2, 2);
    if (wave_on) frags(0, 0, 0, 0, 0, B0);
    int st = 0, gs = 0, r = 0, w3 = 3;              // w3 = (q + 3) mod nq1: the W half tile issued in iteration q
    // Round 6: the fragment ADDRESSES are computed one batch ahead of the reads (ta / tb: the reads at the top of the next iteration, under
    // the second MFMA batch; na / nb: the reads behind the barrier, under the first), so that only the ds_reads themselves sit between
    // two MFMA batches: 3-5 % on every level-0 / level-1 shape (profiles/round6/stencil_dma_late_r6.log; requesting the DMA behind the
    // second batch instead of in front of it bought 2-3 % alone and nothing on top of this).  stagger & 2 (A/B): the round-5 placement
    const bool addr_late = (stagger & 2) != 0;
    unsigned ta[MI], tb, na[MI], nb;
    if (!addr_late) frag_addr(0, 0, 0, 0, 1, ta, tb);
    for (int q = 0; q < nq; ++q) {
        // next half tile's coordinates
        int r1 = r + 1, gs1 = gs;
        if (r1 == IPS) { r1 = 0; ++gs1; }
        const int st1 = (st == 2) ? 0 : st + 1;
        if (wave_on) {
            if (addr_late) frags(gs & 1, r >> 1, r & 1, st, 1, B1);
            else frag_read(ta, tb, B1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");   // buffer 0 (the older reads) is in
            __builtin_amdgcn_sched_barrier(0);
            mfmas(B0);
            __builtin_amdgcn_sched_barrier(0);
            if (!addr_late) frag_addr(gs1 & 1, r1 >> 1, r1 & 1, st1, 0, na, nb);
        }
        // (the halo piece's offset arithmetic — a division, the image-edge tests — under the batch too)
        const bool halo_on = r < H_IT && gs + 1 < ns_tot && wave + NW * r >= pc_lo && wave + NW * r < pc_hi && wave + NW * r < HBLK;
        unsigned hoff = 0;
        if (halo_on && !addr_late) hoff = halo_off(wave + NW * r);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < nq) wait_all_but_last_w();                 // in flight: W(q+1), [halo piece, W(q+2)]
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave has read everything it needs of stage st
        __builtin_amdgcn_s_barrier();
        // halo buffer (gs+1)&1 was last read in slice gs-1; stage st by half tile q (all waves are past their reads of it)
        if (halo_on) {
            if (addr_late) issue_halo(slice_plane(gs + 1), slice_cc(gs + 1), (gs + 1) & 1, wave + NW * r);
            else request_halo(slice_plane(gs + 1), slice_cc(gs + 1), (gs + 1) & 1, wave + NW * r, hoff);
        }
        if (q + 3 < nq) issue_w(w3, st);
        w3 = (w3 + 1 == nq1) ? 0 : w3 + 1;
        if (wave_on) {
            if (q + 1 < nq) {
                if (addr_late) frags(gs1 & 1, r1 >> 1, r1 & 1, st1, 0, B0);
                else frag_read(na, nb, B0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(B1);
            __builtin_amdgcn_sched_barrier(0);
            if (!addr_late) frag_addr(gs1 & 1, r1 >> 1, r1 & 1, st1, 1, ta, tb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (A_lo && q + 1 == nq1) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] *= LO_INV;
        }
        st = st1; r = r1; gs = gs1;
    }
    }
    __syncthreads();                            // every wave is done with the operand buffers
    if (!wave_on) return;                       // rows of another workgroup (tail split)

    // ------------------------------ epilogue ------------------------------
    if constexpr (EPI == E_O32) {
        if (!(stagger & 4) && p.act == PNC_ACT_NONE && n0 + BN <= p.N) {
            // fp32 output straight from the accumulators (round 6): lane (column c, half h) of a 32x32 block holds rows 8 q + 4 h + e of column
            // c, so one store instruction writes two whole 128-byte row pieces — no LDS round trip (160 four-byte staging writes + 40 reads
            // per wave), same bytes to memory.  1.4-3.3 % per launch (profiles/round6/stencil_direct_epilogue_r6.log); acc + bias as in
            // epi_fast: bit-identical.  stagger & 4 (A/B, PNC_OPT_GEMM_FUSE_LN + 2): the staged epilogue
            epi_direct_o32<MI, NI, false>(p, acc, lane, RowHalo<TWS>{base_m, p.Wout, wm * 64}, n0 + wn * (NI * 32));
            return;
        }
    }
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EPITCH);
    epi_fast<MI, NI, EPI>(p, acc, ep, lane, RowHalo<TWS>{base_m, p.Wout, wm * 64}, n0 + wn * (NI * 32), p.N);
}

template <int TWS, int NI>
constexpr int stencil_lds_bytes() {
    constexpr int TW = 1 << TWS, TH = 256 / TW, HROWS = (TH + 2) * (TW + 2);
    return 2 * ((HROWS + 7) / 8) * 1024 + 3 * NI * 64 * 64;
}

template <int TWS, int NI, unsigned EPI>
static int launch_stencil(const PncGemmParams& p, hipStream_t st) {
    constexpr int TW = 1 << TWS, TH = 256 / TW, BN = NI * 64;
    constexpr int lds = stencil_lds_bytes<TWS, NI>();
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = stencil_tile_kernel<TWS, NI, EPI>;
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    const int tiles_m = (p.M / (p.Hout * p.Wout)) * (p.Hout / TH) * (p.Wout >> TWS), tiles_n = (p.N + BN - 1) / BN;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? gopt : (tiles_n > 8 ? 4 : 0);
    if (group_m > tiles_m) group_m = tiles_m;
    if (group_m == 1 || tiles_n < 2) group_m = 0;
    int nfull = tiles_m * tiles_n, tail_f = 1;
    tail_split<256, 4, lds>(tiles_m * tiles_n, nfull, tail_f);            // one workgroup per CU: 256 slots per round
    hipLaunchKernelGGL(kern, dim3(nfull + (tiles_m * tiles_n - nfull) * tail_f), dim3(512), lds, st, p, group_m, nfull, tail_f,
                       pnc_get_option(PNC_OPT_GEMM_STAGGER) == 1 ? 1 : (((pnc_get_option(PNC_OPT_STENCIL_TILES) & 4) ? 2 : 0) | ((pnc_get_option(PNC_OPT_GEMM_FUSE_LN) & 2) ? 4 : 0)));
    // (stagger = 1: the staggered schedule, measured 5-15 % slower than the pipeline, only on request; + 2: the pipeline with the fragment
    // addresses computed next to the reads, as in round 5; + 4: the fp32-only epilogue staged through LDS, as in round 5)
    return pnc_launch_status();
}

template <int TWS, int NI>
static int dispatch_epi(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    switch (epi) {
        case E_O16: return launch_stencil<TWS, NI, E_O16>(p, st);
        case E_O32: return launch_stencil<TWS, NI, E_O32>(p, st);
        case E_O32 | E_O16: return launch_stencil<TWS, NI, E_O32 | E_O16>(p, st);
        case E_R1 | E_O32: return launch_stencil<TWS, NI, E_R1 | E_O32>(p, st);
        case E_R1 | E_O32 | E_O16: return launch_stencil<TWS, NI, E_R1 | E_O32 | E_O16>(p, st);
        default: return PNC_EINVAL;
    }
}

// Geometry code the tile kernel would use for this problem — (TWS << 4) | NI — or 0 when the per-tap gather serves it: stride 2 /
// nearest-x2 gathers, narrow or ragged channel counts, images that do not tile, ragged epilogues, and (unless
// PNC_OPT_STENCIL_TILES = 2: tests) grids of fewer than 160 tiles: the per-tap kernels have split K for those (level 2, 192 tiles:
// +8 % on the tile kernel; a sparse last round — level 1: 384 tiles = 1.5 rounds — is tail-split here as there).
int conv3x3_tile_geometry(const PncGemmParams& p, unsigned epi) {
    const int opt = pnc_get_option(PNC_OPT_STENCIL_TILES) & 3;   // 0 off, 1 auto, 2 wherever the shape allows
    if (!opt) return 0;
    if (p.stride != 1 || p.upsample || p.conv_pad_br || (p.Cin & 63) || p.Hin != p.Hout || p.Win != p.Wout) return 0;
    if (p.K != 9 * p.Cin || p.M % (p.Hout * p.Wout)) return 0;
    if (p.A_lo && p.a_lo_fmt != PNC_LO_F16) return 0;            // the tile kernel's lo pass reads fp16 planes (same MFMA as the hi pass)
    if (epi != E_O16 && epi != E_O32 && epi != (E_O32 | E_O16) && epi != (E_R1 | E_O32) && epi != (E_R1 | E_O32 | E_O16)) return 0;
    int tws = 0;
    if ((p.Hout % 16) == 0 && (p.Wout % 16) == 0) tws = 4;
    else if ((p.Hout % 8) == 0 && (p.Wout % 32) == 0) tws = 5;
    if (!tws) return 0;
    int ni = (p.N % 320 == 0) ? 5 : 4;
    if (ni == 5 && (p.N & 255) == 0) {
        // a grid that fits one round either way: 256-column tiles fill more of the chip (level 2: 48 x 5 = 240 workgroups of
        // 256x256 instead of 48 x 4 = 192 of 256x320)
        const long t5 = (long)(p.M / 256) * (p.N / 320), t4 = (long)(p.M / 256) * (p.N / 256);
        if (t5 < 256 && t4 <= 256) ni = 4;
    }
    if (opt == 1) {
        const long tiles = (long)(p.M / 256) * ((p.N + ni * 64 - 1) / (ni * 64));
        if (tiles < 160 || p.N < 256) return 0;
    }
    return (tws << 4) | ni;
}

int dispatch_conv3x3_tiles(const PncGemmParams& p, unsigned epi, int geometry, hipStream_t st) {
    switch (geometry) {
        case (4 << 4) | 5: return dispatch_epi<4, 5>(p, epi, st);
        case (4 << 4) | 4: return dispatch_epi<4, 4>(p, epi, st);
        case (5 << 4) | 5: return dispatch_epi<5, 5>(p, epi, st);
        case (5 << 4) | 4: return dispatch_epi<5, 4>(p, epi, st);
        default: return PNC_EINVAL;
    }
}

}  // namespace pnc_gemm

PNC_DEFINE_TU_COLLECT(gemm_stencil_tile)
