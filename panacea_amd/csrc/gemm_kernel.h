// gemm_kernel.h — MFMA (v_mfma_f32_32x32x16_f16) GEMM family for gfx950: kernel template + launcher.
//
//   C[M,N] = gatherA[M,K] x W[N,K]^T   fp16 operands, fp32 accumulate, fused epilogue.
//
// One kernel template covers every dense contraction of the Panacea denoising path (include/panacea_hip.h §1):
//   PNC_A_PLAIN     Linear / 1x1 conv on channels-last tokens            (gemm_plain.hip)
//   PNC_A_CONV3X3   implicit-GEMM 3x3 conv over an NHWC image            (gemm_conv3x3.hip)
//   PNC_A_CONV1D_T  temporal k=3 conv over the frames of one pixel       (gemm_conv1d.hip)
//
// Tile: BM x BN block, BK = 64, WGM x WGN waves, each wave owns MI x NI blocks of 32x32.  Operands go HBM -> LDS
// directly (buffer_load_dwordx4 ... lds through a per-tile buffer resource: 32-bit lane offsets, the K tile as scalar offset),
// 16-B chunks in XOR-swizzled 128-B rows; the LDS image of the DMA is lane-linear, so the swizzle is applied to the per-lane
// SOURCE offset and again on the ds_read side.  Out-of-range chunks (conv padding, K/M/N tails) carry the offset
// PNC_BUF_OOB, which the resource's bound turns into zeros without a memory access.
//
// The EPILOGUE is a compile-time parameter (EPI bit set): which fp32 streams are added (res1, res2, row bias), which
// outputs are written (fp32, fp16, channel-major fp16 "V^T"), GEGLU.  The fast variants assume the vector contract
// checked by epi_fast_ok() on the host (N % 8 == 0, 16-byte aligned pointers and leading dimensions), so they contain
// no per-element predicates: every global access is a 16-byte vector on 8 consecutive columns of one row, staged
// through a wave-private LDS region.  Anything else runs E_GENERIC (scalar, predicated, slow, any shape).
// Round 1 carried every option as a run-time branch inside fully unrolled loops: 190 KB of code and 78 spilled VGPRs in
// the 256x320 kernel; the specialised variants are 10-25 KB with no scratch
// (profiles/round2/gemm_codeobj_r2.txt).
//
// "Precise" operands (A_lo != NULL): A = A_hi + 2^-11 * A_lo with both planes fp16.  The K loop first runs over the lo
// plane, scales the accumulators by 2^-11 (exact), then runs over the hi plane: a 22-bit activation operand at twice
// the MFMA work, same tile machinery.
#pragma once
#include "common.h"
#include <atomic>
#include <utility>

namespace pnc_gemm {

constexpr int BK = 64;          // fp16 elements per K tile = 128 B per LDS row
constexpr int BK8 = 128;        // e4m3 elements per K tile of an fp8 lo pass: the same 128-byte rows
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int E8M0_LO_INV = 127 - 11;      // A scale of the fp8 lo pass: 2^-11 (PncGemmParams.A_lo)

enum : unsigned {
    E_R1 = 1,        // += res1 (fp32 stream, may alias out32)
    E_R2 = 2,        // += res2
    E_RB = 4,        // += rowbias[(m / rb_rows) % rb_mod]
    E_O32 = 8,       // fp32 output
    E_O16 = 16,      // fp16 output (+ optional lo plane)
    E_VT = 32,       // column blocks >= n_split go channel-major (V^T)
    E_GEGLU = 64,    // value * gelu(gate) on interleaved 32-column blocks, fp16 output
    E_GENERIC = 128, // run-time flags, scalar predicated accesses: ragged N, unaligned pointers / leading dimensions
    E_GELU = 256,    // erf GELU on the fp16 output (text-tower MLP); its own variant: erff is ~60 instructions per element
    E_LN = 512,      // LayerNorm of the fp32 output rows written as fp16 by the same workgroup (it owns whole rows)
    E_GS = 1024      // GroupNorm(32) statistics of the fp32 output: one {n, mean, M2} record per (64-row wave block, group) (PncGemmParams.gn_part)
};

constexpr float LO_SCALE = 2048.0f, LO_INV = 1.0f / 2048.0f;   // lo plane = (v - hi) * 2^11

// compile-time loop: the index reaches the body as a constant, so accumulator arrays are always indexed statically
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct RowState {               // per staged A row, fixed over the K loop
    int rel;                    // element offset of the row origin from the tile's WINDOW origin (a_window_origin)
    int y, x;                   // conv3x3: output pixel, the row's frame in the upper 16 bits of .x (indexes a view band's column block);
                                // conv1d: t in .y
    bool valid;
};

// The operands reach LDS through buffer resources (common.h: glds16_buf): base = a per-tile window origin in SGPRs, per-lane
// 32-bit byte offsets.  The window keeps every offset of a tile far below 2^31 whatever the tensor size: plain rows start at
// the tile's first row, conv3x3 at the frame of the tile's first pixel, the temporal conv one frame before the tile's first row.
template <int AMODE>
__device__ __forceinline__ int64_t a_window_origin(const PncGemmParams& p, int m0) {
    if (AMODE == PNC_A_PLAIN) return (int64_t)m0 * p.lda;
    if (AMODE == PNC_A_CONV3X3) return (int64_t)(m0 / (p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin;
    // t_halo: A holds T + 2 frames per sample (a halo frame either side of the T the rows speak of): the row of (b, t, pixel) is
    // m + (2 b + 1) Npix, and one frame before the tile's first row always exists
    if (p.t_halo) return (int64_t)(m0 + 2 * ((m0 / p.Npix) / p.T) * p.Npix) * p.Cin;
    return (int64_t)max(0, m0 - p.Npix) * p.Cin;
}

template <int AMODE>
__device__ __forceinline__ RowState make_row(const PncGemmParams& p, int m, int m0) {
    RowState s;
    s.valid = m < p.M;
    const int mm = s.valid ? m : m0;
    if (AMODE == PNC_A_PLAIN) {
        s.rel = (mm - m0) * p.lda; s.y = 0; s.x = 0;
    } else if (AMODE == PNC_A_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int f = mm / hw, pix = mm - f * hw;
        s.y = pix / p.Wout; s.x = (pix - s.y * p.Wout) | (f << 16);
        s.rel = (f - m0 / hw) * p.Hin * p.Win * p.Cin;
    } else {
        const int f = mm / p.Npix;
        s.y = f % p.T; s.x = 0;
        if (p.t_halo) s.rel = (mm - m0 + (2 * (f / p.T - (m0 / p.Npix) / p.T) + 1) * p.Npix) * p.Cin;
        else s.rel = (mm - max(0, m0 - p.Npix)) * p.Cin;
    }
    return s;
}


// GEGLU gate: Phi(g) = (1 + erf(g / sqrt 2)) / 2 tabulated on [-8, 8) in steps of 1/128 as {Phi(x_i), Phi(x_{i+1}) - Phi(x_i)}
// (16 KB, copied into LDS by the GEGLU GEMMs).  Linear interpolation error <= h^2/8 max|Phi''| = 1.8e-6 — 250x below
// the fp16 rounding of the product it feeds — for 9 VALU + one ds_read_b64 per gate instead of ~14 VALU incl. exp + rcp.
constexpr int PHI_N = 2048;
constexpr float PHI_SCALE = 128.0f, PHI_X0 = -8.0f;
constexpr int PHI_BYTES = PHI_N * 8;

__device__ __forceinline__ float gelu_tab_f(float g, const float* tab) {
    float t = fmaf(g, PHI_SCALE, -PHI_X0 * PHI_SCALE);
    t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)PHI_N - 0.001f);
    const int i = (int)t;
    const float f = t - (float)i;
    const float2 e = *reinterpret_cast<const float2*>(tab + 2 * i);
    return g * fmaf(f, e.y, e.x);
}

// byte offset (from the window origin) of the 16-byte chunk (row state s, k index kc) of an A plane, or PNC_BUF_OOB (reads as zero).
// ESZ = bytes per element: 2 (fp16 planes: 8 channels per chunk), 1 (e4m3 lo plane: 16 channels per chunk, Cin % 64 == 0)
template <int AMODE, unsigned ESZ = 2u>
__device__ __forceinline__ unsigned a_chunk_off(const PncGemmParams& p, const RowState& s, int kc) {
    if (!s.valid || kc >= p.K) return PNC_BUF_OOB;
    if (AMODE == PNC_A_PLAIN) {
        return (unsigned)(s.rel + kc) * ESZ;
    } else if (AMODE == PNC_A_CONV3X3) {
        // K order: (ky,kx,ci) for narrow inputs; (ci/64, ky, kx, ci%64) when Cin % 64 == 0, so that the nine tap
        // reads of one 64-channel slice of a pixel neighbourhood are consecutive K tiles and hit L1/L2
        int tap, ci;
        if ((p.Cin & 63) == 0) {
            const int cc = kc / 576, r = kc - cc * 576;
            tap = r >> 6; ci = (cc << 6) + (r & 63);
        } else {
            tap = kc / p.Cin; ci = kc - tap * p.Cin;
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        // x_halo_off: the image is a BAND of a wider one and its columns -1 and Win live in a block [2][frames][Hin][Cin] at A +
        // x_halo_off (engine.ViewShard: the neighbour ranks' edge columns, zeros at the ends of the panorama)
        const int xh = p.x_halo_off != 0 ? 1 : 0;
        const int sx = s.x & 0xFFFF;
        int iy, ix; bool ok;
        if (p.upsample) {
            const int uy = s.y + ky - 1, ux = sx + kx - 1;
            ok = (uy >= 0) && (uy < p.Hout) && (ux >= -xh) && (ux < p.Wout + xh);
            iy = uy >> 1; ix = ux >> 1;                      // ux = -1 -> column -1, ux = Wout -> column Win
        } else {
            const int pad = p.conv_pad_br ? 0 : 1;
            iy = s.y * p.stride + ky - pad; ix = sx * p.stride + kx - pad;
            ok = (iy >= 0) && (iy < p.Hin) && (ix >= -xh) && (ix < p.Win + xh);
        }
        // (32-bit and branch-free: pnc_gemm_f16 admits a column block only when band + block stay below 2^30 elements)
        int off = (iy * p.Win + ix) * p.Cin;
        if (xh) {
            const int f = s.x >> 16, frames = p.M / (p.Hout * p.Wout);
            const int hoff = (int)p.x_halo_off - f * (p.Hin * p.Win * p.Cin) + (((ix < 0 ? 0 : frames) + f) * p.Hin + iy) * p.Cin;
            off = (ix < 0 || ix >= p.Win) ? hoff : off;
        }
        return ok ? (unsigned)(s.rel + off + ci) * ESZ : PNC_BUF_OOB;
    } else {
        // K order: (dt, ci); (ci/64, dt, ci%64) when Cin % 64 == 0 (the three taps of a slice are consecutive K tiles)
        int tap, ci;
        if ((p.Cin & 63) == 0) {
            const int cc = kc / 192, r = kc - cc * 192;
            tap = r >> 6; ci = (cc << 6) + (r & 63);
        } else {
            tap = kc / p.Cin; ci = kc - tap * p.Cin;
        }
        const int tt = s.y + tap - 1;      // (t_halo: frames -1 and T exist in A — the neighbour rank's, or zeros at the clip's ends)
        return (!p.t_halo && (tt < 0 || tt >= p.T)) ? PNC_BUF_OOB : (unsigned)(s.rel + (tap - 1) * p.Npix * p.Cin + ci) * ESZ;
    }
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// exact (erf) GELU of the plain activation epilogue (text-tower MLP: a few thousand rows per sample, not a hot site)
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// the same out of line, for the generic epilogue / split-K reduce (keeps the erff expansion out of their unrolled loops)
__device__ __attribute__((noinline)) static float gelu_erf_call(float v) { return gelu_erf_f(v); }

// fp16 store of 8 consecutive columns, plus the lo plane of a precise operand (fp16 or e4m3: lo_fmt) when the caller asked for one
__device__ __forceinline__ void store_h8(half_t* out16, void* out16_lo, int64_t off, const float (&v)[8], int lo_fmt = PNC_LO_F16) {
    half8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
    *reinterpret_cast<half8v*>(out16 + off) = o;
    if (out16_lo) {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (v[e] - (float)o[e]) * LO_SCALE;
        if (lo_fmt == PNC_LO_E4M3) {
            uint2 w;
            w.x = pack4_e4m3(r[0], r[1], r[2], r[3]);
            w.y = pack4_e4m3(r[4], r[5], r[6], r[7]);
            *reinterpret_cast<uint2*>(reinterpret_cast<char*>(out16_lo) + off) = w;
        } else {
            half8v l;
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = (half_t)r[e];
            *reinterpret_cast<half8v*>(reinterpret_cast<half_t*>(out16_lo) + off) = l;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast row-major epilogue of one wave tile (MI x NI blocks of 32x32), compile-time option set.
//
// Each 32-row x (32|64)-column slab goes through a wave-private LDS region so that a lane ends up with 8 CONSECUTIVE
// columns of one row: 16-byte fp16 stores, two 16-byte fp32 loads/stores.  A 64-column slab uses 8 lanes per row
// (4 passes of 8 rows), the odd 32-column slab of NI = 5 uses 4 lanes per row (2 passes of 16 rows): no idle lanes.
// LDS operations of one wave execute in order, so only lgkmcnt waits separate the phases — no workgroup barrier.
//
// Added fp32 streams (X = res1 | rowbias, Y = res2 | rowbias next to res1) are prefetched ROLLING: as soon as pass ps of
// slab s has consumed its values, the same registers receive the loads of pass ps of slab s+1 — issued before the
// stores of pass ps, so they travel with those stores and under the LDS staging of slab s+1.  In place (res1 == out32)
// a load may not move above an earlier store; rows/columns of different slabs are disjoint, so this order is safe.
// The loads of slab 0 go out after its accumulators have been staged (32 registers free again): with two streams in
// flight the live set stays below 256 VGPRs next to the 160 accumulator registers of the 256x320 tile.
//
// E_LN (the workgroup owns whole rows: N <= BN, two waves per row): every pass also reduces sum / sum of squares of its
// row segment over the lanes of the row and accumulates them per row in LDS; after a workgroup barrier each wave adds its
// partner's half and writes the normalised fp16 row from the values it kept ON CHIP: pass 1 writes every final value back
// into its staging slab and returns the slab to the accumulator registers (dead by then) in the MFMA layout; pass 2 stages
// them again.  (Re-reading the rows from global memory instead was measured: no gain — a 327 KB tile per workgroup does
// not stay in L2, so the re-read costs what the LayerNorm launch's read cost.)  This replaces the LayerNorm launch that
// would otherwise read the stream again from HBM (attention.py:726-747: every residual GEMM of a block is followed by a norm).
// wave-tile row -> output row m.  RowLinear: the GEMM's rows are consecutive; RowHalo (gemm_stencil_tile.hip): the wave tile is
// a strip of a TH x 2^TWS spatial tile of one frame.
struct RowLinear {
    int mw;
    __device__ __forceinline__ int operator()(int lr) const { return mw + lr; }
};
template <int TWS>
struct RowHalo {
    int base, W, lr0;           // m of the tile's first pixel, image width, first tile-local row of this wave
    __device__ __forceinline__ int operator()(int lr) const {
        const int R = lr0 + lr;
        return base + (R >> TWS) * W + (R & ((1 << TWS) - 1));
    }
};

template <int MI, int NI, unsigned EPI, class RM = RowLinear>
__device__ __forceinline__ void epi_fast(const PncGemmParams& p, f32x16 (&acc)[MI][NI], float* ep, int lane,
                                         RM rmap, int nw, int ncols, float2* ln_mine = nullptr,
                                         const float2* ln_partner = nullptr, float* gs_tab = nullptr) {
    constexpr bool R1 = (EPI & E_R1) != 0, R2 = (EPI & E_R2) != 0, RB = (EPI & E_RB) != 0;
    constexpr bool O32 = (EPI & E_O32) != 0, O16 = (EPI & E_O16) != 0, LN = (EPI & E_LN) != 0;
    // E_GS: GroupNorm statistics of the values this wave writes (its MI*32 rows x NI*32 columns; the host admits the variant only
    // when every group of N/32 channels lies inside one wave's columns, cpg even, and the rows are whole and of one frame).  Per
    // slab a lane owns 8 columns = at most two groups' pieces: pair sums over the slab's passes in registers, split at the
    // group boundary once per slab, added over the slab's row lanes (xor shuffles) and kept per 8-column chunk in gs_tab
    // ([NI*4][4] floats of this wave: {sum, squares} of the piece in the chunk's first group, then of the next group's).
    // No atomics and a fixed order everywhere: the records are reproducible bit for bit.
    constexpr bool GS = (EPI & E_GS) != 0;
    static_assert(!GS || (O32 && !LN), "statistics of the fp32 output; not next to the fused LayerNorm");
    const int cpg = GS ? (p.N >> 5) : 1;
    if constexpr (GS) {
        gs_tab[lane] = 0.0f;
        if (lane + 64 < NI * 16) gs_tab[lane + 64] = 0.0f;
    }
    static_assert(!LN || O32, "the fused LayerNorm normalises the fp32 output it has just written");
    constexpr bool HAS_X = R1 || RB, HAS_Y = R1 && (R2 || RB);
    static_assert(!(R2 && !R1), "a single residual is passed as res1");
    static_assert(!(R1 && R2 && RB), "three added streams run the generic epilogue");
    constexpr int ENI = NI < 2 ? NI : 2;
    constexpr int EPITCH = ENI * 32 + 4;
    constexpr int NJ = (NI + ENI - 1) / ENI, NS = NJ * MI;
    constexpr int NPMAX = ENI == 2 ? 4 : 2;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    void* out16_lo = p.out16_lo;
    constexpr bool GELU = (EPI & E_GELU) != 0;
    const bool silu = (!HAS_X) && (p.act == PNC_ACT_SILU);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // (Measured and not kept, round 3: for fp16-only outputs — QKV, q-projections — applying the bias on the accumulators and
    // staging only the fp16 result, as epi_geglu does, is 5-10 % SLOWER than this path (L0 QKV 224 -> 245-250 us): with no
    // arithmetic to take off the LDS round trip, the 16 two-byte staging writes and 16 scalar conversions per block cost more
    // than 16 four-byte writes and 8 packed conversions.  profiles/round3/kbench_r3l_register_epilogues_ab.log)
    f32x4 x0[NPMAX], x1[NPMAX], y0[NPMAX], y1[NPMAX];
    if constexpr (LN) {
        if (lane < MI * 32) ln_mine[lane] = make_float2(0.0f, 0.0f);         // MI * 32 <= 64 rows per wave
    }

    // slab s -> (row block i, first column block jc, width cw in blocks, lanes per row, rows per pass, passes)
    auto load_xy = [&](auto s_, auto ps_) {
        constexpr int s = decltype(s_)::value, ps = decltype(ps_)::value;
        constexpr int jc = (s / MI) * ENI, i = s % MI, cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        constexpr int CPL = cw * 4, RPP = 64 / CPL;
        const int cl = lane % CPL, rl = lane / CPL;
        const int ncol = nw + jc * 32 + cl * 8;
        const int m = rmap(i * 32 + ps * RPP + rl);
        const bool on = (m < p.M) && (ncol < ncols);
        if constexpr (HAS_X) {
            x0[ps] = z4; x1[ps] = z4;
            if (on) {
                const float* xp = R1 ? p.res1 + (int64_t)m * p.ldr1 + ncol
                                     : p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
                x0[ps] = ld4(xp); x1[ps] = ld4(xp + 4);
            }
        }
        if constexpr (HAS_Y) {
            y0[ps] = z4; y1[ps] = z4;
            if (on) {
                const float* yp = R2 ? p.res2 + (int64_t)m * p.ldr2 + ncol
                                     : p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
                y0[ps] = ld4(yp); y1[ps] = ld4(yp + 4);
            }
        }
    };

    static_for<NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value, jc = (s / MI) * ENI, i = s % MI;
        constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        constexpr int CPL = cw * 4, RPP = 64 / CPL, NP = 32 / RPP;
        const int cl = lane % CPL, rl = lane / CPL;
        const int ncol = nw + jc * 32 + cl * 8;
        const bool col_on = ncol < ncols;
        f32x4 b0 = z4, b1 = z4;
        if (p.bias && col_on) { b0 = ld4(p.bias + ncol); b1 = ld4(p.bias + ncol + 4); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // previous slab fully read back from LDS
        float gP[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gQ[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // E_GS: column-pair sums over this slab's passes
        static_for<cw>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)] = acc[i][jc + j][r];
        });
        if constexpr (s == 0 && (HAS_X || HAS_Y)) {
            static_for<NP>([&](auto ps_) { load_xy(std::integral_constant<int, 0>{}, ps_); });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<NP>([&](auto ps_) {
            constexpr int ps = decltype(ps_)::value;
            const float* src = ep + (ps * RPP + rl) * EPITCH + cl * 8;
            const f32x4 a0 = ld4(src), a1 = ld4(src + 4);
            const int m = rmap(i * 32 + ps * RPP + rl);
            const bool on = col_on && m < p.M;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[e] + b0[e]; v[e + 4] = a1[e] + b1[e]; }
            if constexpr (RB && HAS_Y) {            // rowbias next to res1: bias + rowbias first (header order)
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += y0[ps][e]; v[e + 4] += y1[ps][e]; }
            }
            if constexpr (HAS_X) {                  // res1, or rowbias when it is the only stream
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += x0[ps][e]; v[e + 4] += x1[ps][e]; }
            } else {
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                }
                if constexpr (GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
                }
            }
            if constexpr (R2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += y0[ps][e]; v[e + 4] += y1[ps][e]; }
            }
            // rolling prefetch: these registers are free now; the loads go out before this pass's stores
            if constexpr ((HAS_X || HAS_Y) && s + 1 < NS) {
                constexpr int jn = ((s + 1) / MI) * ENI, cwn = (NI - jn) < ENI ? (NI - jn) : ENI;
                constexpr int NPn = 32 / (64 / (cwn * 4));
                if constexpr (ps < NPn) load_xy(std::integral_constant<int, s + 1>{}, ps_);
                // (a narrower next slab has fewer passes; a wider one cannot follow a narrower one: the odd block is last)
            }
            if constexpr (LN) {
                // row statistics of this pass: 8 columns per lane, CPL lanes per row (consecutive lanes)
                float sm = 0.0f, sq = 0.0f;
                if (on) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sm += v[e]; sq = fmaf(v[e], v[e], sq); }
                }
#pragma unroll
                for (int o = 1; o < CPL; o <<= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }
                if (cl == 0) {
                    float2 t = ln_mine[i * 32 + ps * RPP + rl];
                    t.x += sm; t.y += sq;
                    ln_mine[i * 32 + ps * RPP + rl] = t;
                }
                // the final values go back into the slab they came from: after the slab's passes they return to the (dead)
                // accumulator registers in the MFMA layout, so that the normalising pass needs no global re-read — a
                // workgroup's 327 KB tile does not survive in the 4 MB L2 it shares with 31 other CUs
                float* dst = ep + (ps * RPP + rl) * EPITCH + cl * 8;
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if constexpr (GS) {
                if (on) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gP[k] += v[2 * k] + v[2 * k + 1];
                        gQ[k] += fmaf(v[2 * k], v[2 * k], v[2 * k + 1] * v[2 * k + 1]);
                    }
                }
            }
            if (!on) return;
            if constexpr (O32) {
                float* op = p.out32 + (int64_t)m * p.ldc32 + ncol;
                const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                *reinterpret_cast<f32x4*>(op) = o0;
                *reinterpret_cast<f32x4*>(op + 4) = o1;
            }
            if constexpr (O16) store_h8(out16, out16_lo, (int64_t)m * p.ldc16 + ncol, v, p.out_lo_fmt);
        });
        if constexpr (GS) {
            const int nA = min(8, (ncol / cpg + 1) * cpg - ncol);           // columns of this lane's chunk in its first group (even)
            float sA = 0.0f, qA = 0.0f, sB = 0.0f, qB = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool a = 2 * k < nA;
                sA += a ? gP[k] : 0.0f; qA += a ? gQ[k] : 0.0f;
                sB += a ? 0.0f : gP[k]; qB += a ? 0.0f : gQ[k];
            }
#pragma unroll
            for (int o = CPL; o < 64; o <<= 1) {
                sA += __shfl_xor(sA, o, 64); qA += __shfl_xor(qA, o, 64);
                sB += __shfl_xor(sB, o, 64); qB += __shfl_xor(qB, o, 64);
            }
            if (lane < CPL) {                   // rl == 0: this chunk's entry (row blocks i = 0 .. MI-1 add up in slab order)
                f32x4* e = reinterpret_cast<f32x4*>(gs_tab + (jc * 4 + cl) * 4);
                f32x4 t = *e;
                t[0] += sA; t[1] += qA; t[2] += sB; t[3] += qB;
                *e = t;
            }
        }
        if constexpr (LN) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every lane's rows are back in the slab
            static_for<cw>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][jc + j][r] = ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)];
            });
        }
    });
    if constexpr (GS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int m_first = rmap(0);
        if (lane < (NI * 32) / cpg && m_first < p.M) {
            const int c0 = (lane * cpg) >> 3, c1 = ((lane + 1) * cpg - 1) >> 3;
            float S = 0.0f, Q = 0.0f;
            for (int c = c0; c <= c1; ++c) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(gs_tab + c * 4);
                const bool first = (c * 8) / cpg == lane;       // the chunk's first group is this one; else its second piece is
                S += first ? t[0] : t[2];
                Q += first ? t[1] : t[3];
            }
            const float n = (float)(MI * 32 * cpg);
            const float mean = S / n;
            const int f = m_first / p.Npix, chunk = (m_first - f * p.Npix) / (MI * 32), nrec = p.Npix / (MI * 32);
            float* o = p.gn_part + ((int64_t)(f * nrec + chunk) * 32 + nw / cpg + lane) * 3;
            o[0] = n; o[1] = mean; o[2] = fmaxf(Q - S * mean, 0.0f);
        }
    }
    if constexpr (LN) {
        __syncthreads();                      // both halves of every row have their statistics in LDS
        half_t* lnout = reinterpret_cast<half_t*>(p.ln_out16);
        const float invn = 1.0f / (float)p.N;
        static_for<NS>([&](auto s_) {
            constexpr int s = decltype(s_)::value, jc = (s / MI) * ENI, i = s % MI;
            constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;
            constexpr int CPL = cw * 4, RPP = 64 / CPL, NP = 32 / RPP;
            const int cl = lane % CPL, rl = lane / CPL;
            const int ncol = nw + jc * 32 + cl * 8;
            const bool col_on = ncol < ncols;
            f32x4 g0 = z4, g1 = z4, h0 = z4, h1 = z4;
            if (col_on) {
                g0 = ld4(p.ln_gamma + ncol); g1 = ld4(p.ln_gamma + ncol + 4);
                h0 = ld4(p.ln_beta + ncol); h1 = ld4(p.ln_beta + ncol + 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // previous slab fully read back from LDS
            static_for<cw>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)] = acc[i][jc + j][r];
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 w0[NP], w1[NP];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const float* src = ep + (ps * RPP + rl) * EPITCH + cl * 8;
                w0[ps] = ld4(src); w1[ps] = ld4(src + 4);
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int r = i * 32 + ps * RPP + rl;
                const int m = rmap(r);
                if (!(col_on && m < p.M)) continue;
                const float2 a = ln_mine[r], b = ln_partner[r];
                const float mean = (a.x + b.x) * invn;
                // E[x^2] - mean^2 with the subtraction's product fused, written out: left to the compiler's contraction the two
                // kernels that hold this epilogue fused different products (1 fp16 ulp apart in 3e-5 of the outputs)
                const float var = fmaxf(fmaf(-mean, mean, (a.y + b.y) * invn), 0.0f);
                const float rs = rsqrtf(var + p.ln_eps);
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = fmaf((w0[ps][e] - mean) * rs, g0[e], h0[e]);
                    y[e + 4] = fmaf((w1[ps][e] - mean) * rs, g1[e], h1[e]);
                }
                store_h8(lnout, nullptr, (int64_t)m * p.ldln + ncol, y);
            }
        });
    }
}

// GEGLU epilogue: the two column blocks of a staged chunk are a value block and its gate block (engine.pk_geglu);
// 4 lanes per row own 8 of the 32 output columns each.
template <int MI, int NI>
__device__ __forceinline__ void epi_geglu(const PncGemmParams& p, f32x16 (&acc)[MI][NI], float* ep, int lane,
                                          int mw, int nw, const float* phi_tab, const float* pre_bias = nullptr) {
    // pre_bias (register path only): bias of this lane's column in each of the NI column blocks, loaded by the caller — the
    // persistent kernel must not wait on a global load here (vmcnt is in order: it would wait for the prefetched K tile too)
    static_assert(NI % 2 == 0, "GEGLU pairs value / gate column blocks inside a wave");
    constexpr int EPITCH = 2 * 32 + 4;
    constexpr int CPL = 4, RPP = 16, NP = 2;
    const int cl = lane % CPL, rl = lane / CPL;
    const int Nout = p.N >> 1;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    void* out16_lo = p.out16_lo;
    if (out16_lo == nullptr && mw + MI * 32 <= p.M && ((nw + NI * 32) >> 1) <= Nout) {
        // (wave tiles inside the matrix and plain fp16 outputs; ragged tiles and lo planes take the staged path below)
        // The value and the gate of an output element sit in the SAME lane and accumulator slot (block jc and jc + 1, column
        // lane & 31), so the gate arithmetic runs on the accumulators where they are; only the fp16 RESULT is staged — for the
        // transposition to 16-byte row segments.  A quarter of the staging bytes of the path below (16 x 2 B written, 2 x 16 B
        // read per lane and slab, instead of 32 x 4 B / 8 x 16 B) and, more to the point, no arithmetic waits on an LDS round
        // trip: every slab of the wave tile has its own 2 KB staging buffer and a wave's DS operations execute in order, so the
        // gate arithmetic of all slabs runs first (phase 1) and the 16-byte reads and stores follow back to back (phase 2).
        // Measured: L0 FF1 517 -> 467-480 us, L1 377-386 -> 360-365 us, L2 -3 % (profiles/round3/kbench_r3l_register_epilogues_ab.log).
        // Same products, same roundings (fp32, then fp16): bit-identical to the staged path.
        constexpr int P16 = 32;                                          // halves per staged row; one 2 KB buffer per slab
        static_assert((NI / 2) * MI * 32 * P16 * 2 <= 32 * EPITCH * 4, "all slabs of a wave tile are staged side by side");
        typedef half_t __attribute__((may_alias)) half_st;               // staged as halves, read back as 16-byte words
        typedef int4 __attribute__((may_alias)) int4_st;
        half_st* st = reinterpret_cast<half_st*>(ep);
        const int c = lane & 31;
        // phase 1: gate arithmetic on the accumulators, results to the staging buffers.  Per slab the 16 table reads are
        // issued together (index / fraction first, reads, then the products) instead of one read -> wait -> use chain each.
        static_for<NI / 2>([&](auto jc_) {
            constexpr int jc = decltype(jc_)::value * 2;
            const float bv = pre_bias ? pre_bias[jc] : (p.bias ? p.bias[nw + jc * 32 + c] : 0.0f);
            const float bg = pre_bias ? pre_bias[jc + 1] : (p.bias ? p.bias[nw + jc * 32 + 32 + c] : 0.0f);
            static_for<MI>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                half_st* sb = st + (decltype(jc_)::value * MI + i) * (32 * P16);
                float gx[16], fr[16];
                int ix[16];
                float2 e[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {                           // = gelu_tab_f(), split around its table read
                    gx[r] = acc[i][jc + 1][r] + bg;
                    float t = fmaf(gx[r], PHI_SCALE, -PHI_X0 * PHI_SCALE);
                    t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)PHI_N - 0.001f);
                    ix[r] = (int)t;
                    fr[r] = t - (float)ix[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = *reinterpret_cast<const float2*>(phi_tab + 2 * ix[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                {
                    // the fp32 product is kept opaque so that hipcc does not fuse multiply + conversion into v_fma_mixlo_f16 (one
                    // rounding): the result is rounded to fp32, then to fp16, exactly like the staged path and the other tiles
                    float prod = (acc[i][jc][r] + bv) * (gx[r] * fmaf(fr[r], e[r].y, e[r].x));
                    asm("" : "+v"(prod));
                    sb[mfma32_row(r, lane) * P16 + c] = (half_t)prod;
                }
            });
        });
        // phase 2: 16-byte row segments out — no per-lane predicates, no branches: all reads go out before the first store
        static_for<NI / 2>([&](auto jc_) {
            constexpr int jc = decltype(jc_)::value * 2;
            const int ncol0 = (nw + jc * 32) >> 1;                       // first output column of the block pair
            static_for<MI>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                const half_st* sb = st + (decltype(jc_)::value * MI + i) * (32 * P16);
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    const int row = ps * RPP + rl;
                    const int4 v = *reinterpret_cast<const int4_st*>(sb + row * P16 + cl * 8);
                    *reinterpret_cast<int4_st*>(out16 + (int64_t)(mw + i * 32 + row) * p.ldc16 + ncol0 + cl * 8) = v;
                }
            });
        });
        return;
    }
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<NI / 2>([&](auto jc_) {
        constexpr int jc = decltype(jc_)::value * 2;
        const int nin = nw + jc * 32 + cl * 8;                          // first input column (N space of W / bias)
        const int ncol = ((nw + jc * 32) >> 1) + cl * 8;                // first output column of this lane
        const bool col_on = ncol < Nout;
        f32x4 b0 = z4, b1 = z4, g0b = z4, g1b = z4;
        if (p.bias && col_on) {
            b0 = ld4(p.bias + nin); b1 = ld4(p.bias + nin + 4);
            g0b = ld4(p.bias + nin + 32); g1b = ld4(p.bias + nin + 36);
        }
        static_for<MI>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // previous slab fully read
            static_for<2>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)] = acc[i][jc + j][r];
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 a0[NP], a1[NP], g0[NP], g1[NP];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const float* src = ep + (ps * RPP + rl) * EPITCH + cl * 8;
                a0[ps] = ld4(src); a1[ps] = ld4(src + 4);
                g0[ps] = ld4(src + 32); g1[ps] = ld4(src + 36);
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int m = mw + i * 32 + ps * RPP + rl;
                if (!col_on || m >= p.M) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (a0[ps][e] + b0[e]) * gelu_tab_f(g0[ps][e] + g0b[e], phi_tab);
                    v[e + 4] = (a1[ps][e] + b1[e]) * gelu_tab_f(g1[ps][e] + g1b[e], phi_tab);
                }
                store_h8(out16, out16_lo, (int64_t)m * p.ldc16 + ncol, v, p.out_lo_fmt);
            }
        });
    });
}

// channel-major ("V^T") output of one wave tile: a lane already holds 4 consecutive rows of one column; rows
// 8*r4 + {0..3} live in lane l, {4..7} in lane l+32 (same column): one xor-32 exchange per pair of r4 gives each lane 8
// consecutive rows -> 16-byte stores along the token axis.  Vector contract: M % 8 == 0, t_rows % 8 == 0.
template <int MI, int NI>
__device__ __forceinline__ void epi_vt(const PncGemmParams& p, f32x16 (&acc)[MI][NI], int lane, int mw, int nw) {
    half_t* out16t = reinterpret_cast<half_t*>(p.out16t);
    const int col = lane & 31, grp = lane >> 5;
    static_for<MI * NI>([&](auto ij_) {
        constexpr int i = decltype(ij_)::value / NI, j = decltype(ij_)::value % NI;
        const int n = nw + j * 32 + col;
        const bool col_on = n < p.N;
        const float bn = (p.bias && col_on) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            half4v he, ho;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                he[q] = (half_t)(acc[i][j][(2 * rp) * 4 + q] + bn);
                ho[q] = (half_t)(acc[i][j][(2 * rp + 1) * 4 + q] + bn);
            }
            union { half4v h; int2 w; } snd, rcv;
            snd.h = grp ? he : ho;
            rcv.w.x = __shfl_xor(snd.w.x, 32, 64);
            rcv.w.y = __shfl_xor(snd.w.y, 32, 64);
            half8v o8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o8[q] = grp ? rcv.h[q] : he[q];
                o8[4 + q] = grp ? ho[q] : rcv.h[q];
            }
            const int m8 = mw + i * 32 + 8 * (2 * rp + grp);            // this lane's 8-row group
            if (col_on && m8 < p.M) {
                const int g = m8 / p.t_rows, t0 = m8 - g * p.t_rows;
                *reinterpret_cast<half8v*>(out16t + (int64_t)g * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + t0) = o8;
            }
        }
    });
}

// DIRECT fp32 epilogue of a full wave tile (round 6): out32 = acc + bias (+ res1), straight from the accumulators.  Lane (column c, half h)
// of a 32x32 block holds rows 8 q + 4 h + e of column c: one four-byte store / load instruction moves two whole 128-byte row pieces, so the
// accesses are as wide as the staged epilogue's at the memory side, and the LDS round trip (160 four-byte staging writes + 40 reads per wave)
// is gone.  The residual of block b + 1 is requested before block b is combined.  Same additions in the same order as epi_fast
// ((acc + bias) + res1): bit-identical.  The caller guarantees: every row / column inside the matrix, no activation.
template <int MI, int NI, bool R1, class RM>
__device__ __forceinline__ void epi_direct_o32(const PncGemmParams& p, f32x16 (&acc)[MI][NI], int lane, RM rmap, int nw) {
    // Addressing through buffer resources: base = the wave tile's first row, per-lane byte offset of (row block i, this lane's half, its
    // column) + the column block as the instruction offset + the row's offset inside the block as the SCALAR offset — no vector
    // arithmetic per access.  rmap is additive over (block, half, row) for both row maps (RowLinear; RowHalo: 4 h + (r & 3) stays inside
    // a 16- / 32-pixel tile row).
    const int col = lane & 31, h4 = 4 * (lane >> 5);
    const int m00 = rmap(0);
    const buffer_rsrc_t ro = make_rsrc(p.out32 + (int64_t)m00 * p.ldc32, 0x7FFFFF00u);
    const buffer_rsrc_t rr = make_rsrc(R1 ? p.res1 + (int64_t)m00 * p.ldr1 : p.out32, 0x7FFFFF00u);
    int vo[MI], vr[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int dm = rmap(i * 32 + h4) - m00;
        vo[i] = (dm * p.ldc32 + nw + col) * 4;
        vr[i] = R1 ? (dm * p.ldr1 + nw + col) * 4 : 0;
    }
    unsigned x[2][16];
    auto load_res = [&](auto b_) {
        constexpr int b = decltype(b_)::value, i = b / NI, j = b % NI;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int so = (rmap((r & 3) + 8 * (r >> 2)) - m00) * p.ldr1 * 4;          // (uniform)
            x[b & 1][r] = __builtin_amdgcn_raw_buffer_load_b32(rr, vr[i] + j * 128, so, 0);
        }
    };
    if constexpr (R1) load_res(std::integral_constant<int, 0>{});
    static_for<MI * NI>([&](auto b_) {
        constexpr int b = decltype(b_)::value, i = b / NI, j = b % NI;
        const float bn = p.bias ? p.bias[nw + j * 32 + col] : 0.0f;
        if constexpr (R1 && b + 1 < MI * NI) load_res(std::integral_constant<int, b + 1>{});
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] + bn;
            if constexpr (R1) v += __builtin_bit_cast(float, x[b & 1][r]);
            const int so = (rmap((r & 3) + 8 * (r >> 2)) - m00) * p.ldc32 * 4;         // (uniform)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, vo[i] + j * 128, so, 0);
        }
    });
}

// Generic epilogue: every option is a run-time flag, every access scalar and predicated.  Correct for any N, leading
// dimension and alignment; used by the few ragged launches of the path (the 4-channel output head) and by callers
// outside the vector contract.  No prefetch arrays: it must not need scratch either.
template <int MI, int NI>
__device__ __forceinline__ void epi_generic(const PncGemmParams& p, f32x16 (&acc)[MI][NI], int lane, int mw, int nw) {
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    half_t* out16_lo = reinterpret_cast<half_t*>(p.out16_lo);
    half_t* out16t = reinterpret_cast<half_t*>(p.out16t);
    const int col = lane & 31;
    static_for<MI * NI>([&](auto ij_) {
        constexpr int i = decltype(ij_)::value / NI, j = decltype(ij_)::value % NI;
        const int n = nw + j * 32 + col;
        if (n >= p.N) return;
        const float bn = p.bias ? p.bias[n] : 0.0f;
        const bool to_t = out16t && n >= p.n_split;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int m = mw + i * 32 + mfma32_row(r, lane);
            if (m >= p.M) continue;
            float v = acc[i][j][r] + bn;
            if (p.rowbias) v += p.rowbias[(int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + n];
            if (p.act == PNC_ACT_SILU) v = silu_f(v);
            if (p.act == PNC_ACT_GELU) v = gelu_erf_call(v);
            if (p.res1) v += p.res1[(int64_t)m * p.ldr1 + n];
            if (p.res2) v += p.res2[(int64_t)m * p.ldr2 + n];
            if (to_t) {
                const int g = m / p.t_rows, t = m - g * p.t_rows;
                out16t[(int64_t)g * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + t] = (half_t)v;
                continue;
            }
            if (p.out32) p.out32[(int64_t)m * p.ldc32 + n] = v;
            if (out16) {
                const half_t h = (half_t)v;
                out16[(int64_t)m * p.ldc16 + n] = h;
                if (out16_lo) {
                    const float r = (v - (float)h) * LO_SCALE;
                    if (p.out_lo_fmt == PNC_LO_E4M3)
                        reinterpret_cast<unsigned char*>(p.out16_lo)[(int64_t)m * p.ldc16 + n] = (unsigned char)(pack4_e4m3(r, 0.0f, 0.0f, 0.0f) & 0xFFu);
                    else out16_lo[(int64_t)m * p.ldc16 + n] = (half_t)r;
                }
            }
        }
    });
}

template <int AMODE, int BM, int BN, int WGM, int WGN, int STAGES, bool PIPE, unsigned EPI>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_glds_kernel(const PncGemmParams pin, const int ksplit,
                                                                   const int nfull, const int tail_f,
                                                                   const float* __restrict__ phi_g, const int group_m,
                                                                   const int stagger_min_in) {
    PncGemmParams p = pin;
    constexpr int NW = WGM * WGN;                          // waves per workgroup
    constexpr int MI = BM / WGM / 32, NI = BN / WGN / 32;
    constexpr int RPI = NW * 8;                            // rows staged per DMA iteration (8 rows per wave)
    constexpr int A_IT = BM / RPI, B_IT = BN / RPI;
    constexpr int LOADS = A_IT + B_IT;                     // DMA instructions per thread per K tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must be a multiple of the DMA row group");
    constexpr int ENI = NI < 2 ? NI : 2;                   // column blocks staged per epilogue pass
    constexpr int EPITCH = ENI * 32 + 4;                   // floats per staged epilogue row
    constexpr bool GEGLU = (EPI & E_GEGLU) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ A_lo = reinterpret_cast<const half_t*>(p.A_lo);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // split K (ksplit > 1): block b = (slice, tile); slice s runs K tiles [s*nt/S, (s+1)*nt/S) and writes its raw fp32
    // accumulators to ws[s][M][N] (the host launches the E_O32 variant with the epilogue options cleared);
    // splitk_reduce_kernel sums the slices in order and applies the epilogue.
    // Tail split (tail_f = 2 or 4): the last (ntile_mn - nfull) output tiles - the partial round that would leave most
    // CUs idle - are each run by tail_f workgroups that own BM / tail_f rows of the tile: the waves of the other row
    // groups skip their MFMAs and epilogue (their A rows are DMA'd as out-of-bounds offsets = zeros), all waves still stage W.  Rows are
    // independent in a GEMM, so the result does not depend on the split.
    const int ntile_mn = tiles_m * tiles_n;
    int kslice = 0, tile, part = 0;
    if (ksplit > 1) {
        const int blk = xcd_remap(blockIdx.x, ntile_mn * ksplit);
        kslice = blk / ntile_mn; tile = blk - kslice * ntile_mn;
    } else if ((int)blockIdx.x < nfull) {
        tile = xcd_remap(blockIdx.x, nfull);
    } else {
        const int j = (int)blockIdx.x - nfull;
        tile = nfull + j / tail_f; part = j - (j / tail_f) * tail_f;
    }
    const bool split_rows = (ksplit == 1) && ((int)blockIdx.x >= nfull) && (tail_f > 1);
    // Tile id -> (tm, tn).  Default: tn fastest, so the ~32 tiles an XCD runs at once are 32 / tiles_n row panels x all
    // column tiles.  With many column tiles (FF1: 10-40, QKV at C = 1280: 12-15) that is ONE panel against the whole of W,
    // and where W exceeds the 4 MB L2 (every level but 0) W is re-streamed from the fabric once per row panel: 1.26 GB per
    // FF1 launch at every level (profiles/round2/pmc_precise_fetch_by_kernel.txt: 52 GB per step in the GEGLU kernel alone).
    // group_m > 0: walk group_m row panels x the column tiles instead (tm fastest inside a group), so the concurrent set
    // is group_m x (32 / group_m) tiles and each W column tile is fetched once per GROUP of panels.  Same tiles, same
    // arithmetic: results are bit-identical.
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
    if constexpr (AMODE == PNC_A_CONV1D_T) {
        // Temporal conv: the three taps of a row panel are the panels of frames t - 1, t, t + 1 at the SAME pixels — with the row
        // panels in memory order (frame-major) the panel of frame t is fetched again, ~48 panels later and mostly on another
        // XCD, for frame t + 1 and t - 1: the operand crossed the fabric three times (profiles/round4: 11.2 GB per step for the
        // first temporal site against 6.2 GB of operands).  Walk the panels FRAME-FASTEST instead (pixel block outer): the
        // consecutive tile ids one XCD works through are the frames of one pixel block, and two of the three reads hit its L2.
        // A permutation of the row panels: same tiles, same arithmetic.
        const int pb_n = p.Npix / BM;                        // row panels per frame
        if (pb_n * BM == p.Npix && pb_n > 1) {
            const int nbt = tiles_m / pb_n;                  // frames (b, t) of the launch
            tm = (tm % nbt) * pb_n + tm / nbt;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int ntiles_all = (p.K + BK - 1) / BK;
    const int kt_begin = (int)((int64_t)kslice * ntiles_all / ksplit);
    const int ntiles = (int)((int64_t)(kslice + 1) * ntiles_all / ksplit) - kt_begin;
    if (ksplit > 1) p.out32 = p.ws + (int64_t)kslice * p.M * p.N;
    // precise operand: the lo plane's K tiles run first.  fp16 lo plane: the same K tiles as the hi plane, then the accumulators
    // are scaled by 2^-11.  e4m3 lo plane (lo8): 128 k per 128-byte LDS row -> half the tiles, DMA pieces and barriers; the
    // block-scaled fp8 MFMA carries the 2^-11 as its A scale and the weight row's exponent as its B scale, so the lo products
    // land in the accumulators at their final weight and the hi pass simply continues.
    const bool lo8 = A_lo && p.a_lo_fmt == PNC_LO_E4M3;
    const int nlo_all = lo8 ? (p.K + BK8 - 1) / BK8 : ntiles_all;
    const int kt_begin_lo = (int)((int64_t)kslice * nlo_all / ksplit);
    const int nt_lo = A_lo ? (int)((int64_t)(kslice + 1) * nlo_all / ksplit) - kt_begin_lo : 0;
    const int ntot = ntiles + nt_lo;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations (M0) and wave-row tests stay on the SALU
    const int wm = wave / WGN, wn = wave % WGN;

    // DMA assignment: lane l of wave w fills slot (l&7) of row i*32 + w*8 + (l>>3); the slot holds the
    // chunk slot ^ ((row>>1)&7), and (row>>1)&7 does not depend on i
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    const int rows_lo = split_rows ? part * (BM / tail_f) : 0;                 // tile-local row range of this workgroup
    const int rows_hi = split_rows ? rows_lo + BM / tail_f : BM;
    const bool wave_on = (wm * (MI * 32) >= rows_lo) && (wm * (MI * 32) < rows_hi);
    RowState rows[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = i * RPI + srow;
        rows[i] = make_row<AMODE>(p, m0 + r, m0);
        rows[i].valid = rows[i].valid && (r >= rows_lo) && (r < rows_hi);
    }
    // buffer resources: the A plane(s) from the tile's window origin, W from the tile's first row.  A row's offset is fixed over
    // the K loop for plain A and for W (the K tile enters as the scalar offset); the gathers recompute theirs per K tile.
    const int64_t a_origin = a_window_origin<AMODE>(p, m0);
    const buffer_rsrc_t rs_a = make_rsrc(A + a_origin, 0x7FFFFF00u);
    const buffer_rsrc_t rs_alo = make_rsrc(lo8 ? static_cast<const void*>(reinterpret_cast<const char*>(p.A_lo) + a_origin)
                                               : static_cast<const void*>((A_lo ? A_lo : A) + a_origin), 0x7FFFFF00u);
    const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);
    const buffer_rsrc_t rs_wlo = make_rsrc(lo8 ? static_cast<const void*>(reinterpret_cast<const char*>(p.W_lo) + (int64_t)n0 * p.ldw_lo)
                                               : static_cast<const void*>(Wt + (int64_t)n0 * p.ldw), 0x7FFFFF00u);
    unsigned woff[B_IT], aoff[A_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int nl = i * RPI + srow;
        woff[i] = (n0 + nl < p.N) ? (unsigned)(nl * p.ldw + schunk * 8) * 2u : PNC_BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
        aoff[i] = (AMODE == PNC_A_PLAIN && rows[i].valid) ? (unsigned)(rows[i].rel + schunk * 8) * 2u : PNC_BUF_OOB;
    const int kt_tail = (p.K & (BK - 1)) ? ntiles_all - 1 : -1;      // the one K tile with chunks beyond K, if any
    // DMA pieces [Q0, Q1) of K tile kt_local into `stage` (pieces 0 .. A_IT-1: the A row groups, A_IT .. LOADS-1: the W row groups;
    // the range is compile-time so that the staggered schedule below can spread a tile's pieces over its phases)
    auto issue_part = [&](int kt_local, int stage, auto q0_, auto q1_) __attribute__((always_inline)) {
        constexpr int Q0 = decltype(q0_)::value, Q1 = decltype(q1_)::value;
        const bool lo = kt_local < nt_lo;
        char* sa = smem + stage * STAGE + wave * 1024;
        char* sb = sa + A_BYTES;
        if (lo && lo8) {                                 // (uniform) e4m3 tile: chunk = 16 k, byte offsets = element offsets
            // The lane offsets of this branch are derived from the hi pass's on the spot.  Opaque copies of the two lane constants
            // keep hipcc from hoisting them out of the K loop as a second set of loop invariants: next to 160 accumulator
            // registers the kernel has ~12 VGPRs to spare, and 9-20 more invariants spilled 100-300 registers (round 3).
            int schunk8 = schunk, srow8 = srow;
            asm volatile("" : "+v"(schunk8), "+v"(srow8));
            const int kt8 = kt_begin_lo + kt_local;
            const int kc8 = kt8 * BK8 + schunk8 * 16;
            const unsigned ks8 = (unsigned)kt8 * BK8;
            const bool k_on = kc8 < p.K;                 // false only in the chunks of the last tile beyond K
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < Q0 || i >= Q1) continue;
                if constexpr (AMODE == PNC_A_PLAIN)      // (rel + 8 schunk) * 2 -> rel + 16 schunk
                    glds16_buf(rs_alo, (k_on && aoff[i] != PNC_BUF_OOB) ? (aoff[i] >> 1) + (unsigned)schunk8 * 8u : PNC_BUF_OOB, ks8,
                               sa + i * (RPI * 128));
                else glds16_buf(rs_alo, a_chunk_off<AMODE, 1u>(p, rows[i], kc8), 0u, sa + i * (RPI * 128));
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                if (A_IT + i < Q0 || A_IT + i >= Q1) continue;
                glds16_buf(rs_wlo, (k_on && woff[i] != PNC_BUF_OOB) ? (unsigned)((i * RPI + srow8) * p.ldw_lo + schunk8 * 16) : PNC_BUF_OOB,
                           ks8, sb + i * (RPI * 128));
            }
            return;
        }
        const int kt = (lo ? kt_begin_lo : kt_begin - nt_lo) + kt_local;
        const buffer_rsrc_t rs = lo ? rs_alo : rs_a;
        const int kc = kt * BK + schunk * 8;
        const unsigned ks = (unsigned)kt * (BK * 2);     // the K tile as the scalar byte offset of plain rows
        if (kt != kt_tail) {                             // (uniform) no per-lane predicate on the K index
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < Q0 || i >= Q1) continue;
                if constexpr (AMODE == PNC_A_PLAIN) glds16_buf(rs, aoff[i], ks, sa + i * (RPI * 128));
                else glds16_buf(rs, a_chunk_off<AMODE>(p, rows[i], kc), 0u, sa + i * (RPI * 128));
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                if (A_IT + i < Q0 || A_IT + i >= Q1) continue;
                glds16_buf(rs_w, woff[i], ks, sb + i * (RPI * 128));
            }
        } else {
            const bool k_on = kc < p.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                if (i < Q0 || i >= Q1) continue;
                if constexpr (AMODE == PNC_A_PLAIN) glds16_buf(rs, k_on ? aoff[i] : PNC_BUF_OOB, ks, sa + i * (RPI * 128));
                else glds16_buf(rs, a_chunk_off<AMODE>(p, rows[i], kc), 0u, sa + i * (RPI * 128));
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                if (A_IT + i < Q0 || A_IT + i >= Q1) continue;
                glds16_buf(rs_w, k_on ? woff[i] : PNC_BUF_OOB, ks, sb + i * (RPI * 128));
            }
        }
    };
    auto issue_tile = [&](int kt_local, int stage) __attribute__((always_inline)) {
        issue_part(kt_local, stage, std::integral_constant<int, 0>{}, std::integral_constant<int, LOADS>{});
    };

    // GEGLU: the Phi table rides into LDS (behind the operand ring) with the first K tile
    constexpr int RING_BYTES = STAGES * STAGE;
    if constexpr (GEGLU) {
        const char* tab = reinterpret_cast<const char*>(phi_g);
#pragma unroll
        for (int c = wave; c < PHI_BYTES / 1024; c += NW)
            glds16(reinterpret_cast<const half_t*>(tab + c * 1024 + lane * 16), smem + RING_BYTES + c * 1024);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fk = lane >> 5;
    auto compute = [&](int stage, int mid_issue = -1) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        if (PIPE) {
            // fragments of k-step ks+1 are read while the MFMAs of k-step ks run (register double buffer)
            half8v af[2][MI], bf[2][NI];
            auto frags = [&](int ks, int b) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[b][i] = *reinterpret_cast<const half8v*>(
                        sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bf[b][j] = *reinterpret_cast<const half8v*>(
                        sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
            };
            frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (ks + 1 < BK / 16) frags(ks + 1, (ks + 1) & 1);
                // keep the reads of k-step ks+1 AHEAD of the MFMAs of k-step ks (hipcc otherwise sinks them behind the
                // MFMAs and then waits lgkmcnt(0) right after issuing them, exposing the LDS latency every k-step)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 1 && mid_issue >= 0) { issue_tile(mid_issue, mid_issue & 1); __builtin_amdgcn_sched_barrier(0); }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                half8v af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *reinterpret_cast<const half8v*>(
                        sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bf[j] = *reinterpret_cast<const half8v*>(
                        sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (ks == 1 && mid_issue >= 0) issue_tile(mid_issue, mid_issue & 1);
            }
        }
    };
    // e4m3 lo tile: two MFMA windows of 64 k (the last tile of K = 320 holds one).  Lane (row r, group g) supplies bytes
    // 32 g .. 32 g + 31 of the window for both operands (element j of a lane group pairs with element j of the same group of the
    // other operand: tools/exp/mx_mfma_probe.hip) = chunks 2g, 2g+1 of the window: two ds_read_b128 per fragment, same swizzle.
    auto compute8 = [&](int stage, int kt_local) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        const int nwin = (p.K - (kt_begin_lo + kt_local) * BK8) > 64 ? 2 : 1;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            if (w < nwin) {
                i32x8 af[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * (MI * 32) + i * 32 + frow;
                    const i32x4 a0 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2));
                    const i32x4 a1 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2 + 1));
                    af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int row = wn * (NI * 32) + j * 32 + frow;
                    const i32x4 b0 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2));
                    const i32x4 b1 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2 + 1));
                    const i32x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf, acc[i][j], 0, 0, 0, E8M0_LO_INV, 0, p.w_lo_exp);
                    // one B fragment (8 registers) in flight: hipcc otherwise hoists the reads of all NI column blocks (40 registers
                    // at NI = 5) above the first MFMA and spills next to the 160 accumulator registers
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    auto scale_lo = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= LO_INV;
    };

    // The second-dispatched half of an 8-wave workgroup loses every issue arbitration by age (MI355X_MICROARCH.md, two waves per
    // SIMD): static priority for it.  Plain-A GEMMs -1.9 ms per step in a same-box A/B; the gathers (+0.3 / +0.4 ms) keep age order.
    // STAGGERED schedule (round 5; 8-wave geometries on two stages, PNC_OPT_GEMM_STAGGER): a K tile is four PHASES, one per k-step —
    //     fragment reads of the k-step + a third of the NEXT tile's DMA pieces | s_barrier | MI x NI MFMAs | s_barrier
    // and waves 4-7 (the second wave of every SIMD) run ONE barrier behind waves 0-3: while one wave of a SIMD is in its MFMA
    // cluster the other reads its fragments and issues its DMA, and the barriers hold that alternation (the two-group form of the
    // HIP guide's 256^2 8-phase template on this kernel's stages).  vmcnt(0) once per K tile, before the first barrier of phase 3 — a
    // whole MFMA cluster after the last DMA issue —, together with lgkmcnt(0): the OTHER group is one barrier away from reading the
    // next tile / overwriting this one.  Same K order and MFMA order per accumulator as the loops below: bit-identical results.
    // Measured.  In a stand-alone probe of this geometry under SUSTAINED load (tools/exp/gemm_phase_probe.hip, back-to-back launches,
    // profiles/round5/gemm_phase_probe_r5a.log, ..._r5c_*.log): +6 .. +19 % on every K >= 640 shape, warm and cold operands alike (L1
    // conv-K 907 -> 1082 TFLOP/s, L2 FF2 1104 -> 1259 = the vendor GEMM's 1259); the group offset is the whole effect (phases without
    // it: -2 %), priority flips around the MFMA clusters are flat, DMA inside the MFMA clusters is 40 % slower, a finer ring of k-half
    // units with counted vmcnt is slower than full-tile stages.  IN THIS KERNEL it does not carry over: the library's launches, timed
    // alone, already run the loops below at 1.14-1.30 PFLOP/s marginal (profiles/round5/stagger_ksweep_r5f.log; the probe's copy of the
    // same loop, throttled by its own sustained load, ran 0.9-1.1), the staggered loop adds +3-4 % of marginal rate where W is wide
    // (N = 1280, operands from L2) and LOSES 16 % where one column tile streams A from HBM (N = 320: its DMA has at most one K tile
    // to land); whole network 157.31 -> 157.15 ms (stagger_ab_whole_network_r5e.log).  So: ON in the persistent GEGLU kernel (FF1
    // at levels 1-2: +5-6 % in the network, +11 % alone — wide N, A from L2, the next output tile's first K tile requested inside
    // the phases), here only on request (PNC_OPT_GEMM_STAGGER = 1: the bit-identity tests and the A/B tools).
    // Where it can run: plain A (the gathers' per-piece address arithmetic sits on the critical path of a phase: the per-tap conv3x3 /
    // temporal conv launches measured 4-20 % SLOWER staggered, profiles/round5/stagger_kbench_r5c_generic_issue_path.log), K a
    // multiple of 64, no fp16 lo plane, and not in the row-split workgroups of a sparse last round (one of the two groups idles there).
    const bool direct_epi = (stagger_min_in & 0x100) != 0;
    const int stagger_min = stagger_min_in & 0xFF;
    bool staggered = false;
    if constexpr (STAGES == 2 && NW == 8 && AMODE == PNC_A_PLAIN)
        staggered = stagger_min == 1 && kt_tail < 0 && (!A_lo || lo8) && !split_rows && ksplit == 1;
    if (AMODE == PNC_A_PLAIN && NW == 8 && wave >= 4 && !staggered) __builtin_amdgcn_s_setprio(1);
    if (staggered) {
        if constexpr (STAGES == 2 && NW == 8 && AMODE == PNC_A_PLAIN) {
            constexpr int Q0 = (LOADS + 2) / 3, Q1 = (LOADS - Q0 + 1) / 2;
            const int grp = wave >> 2;
            half8v af[MI], bf[NI];
            i32x8 af8[MI], bf8[NI];
            auto rd = [&](int stage, int ks) {
                const char* sa = smem + stage * STAGE;
                const char* sb = sa + A_BYTES;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *reinterpret_cast<const half8v*>(sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bf[j] = *reinterpret_cast<const half8v*>(sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
            };
            auto rd8 = [&](int stage, int w) {
                const char* sa = smem + stage * STAGE;
                const char* sb = sa + A_BYTES;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int row = wm * (MI * 32) + i * 32 + frow;
                    const i32x4 a0 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2));
                    const i32x4 a1 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2 + 1));
                    af8[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int row = wn * (NI * 32) + j * 32 + frow;
                    const i32x4 b0 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2));
                    const i32x4 b1 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2 + 1));
                    bf8[j] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            // first barrier of a phase (+ this wave's fragment reads have returned), second barrier
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
            issue_tile(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int n8 = lo8 ? nt_lo : 0;
            if (grp == 1) __builtin_amdgcn_s_barrier();               // group 1 runs one barrier behind group 0 from here on
            for (int kt = 0; kt < n8; ++kt) {                         // e4m3 lo tiles: one phase per 64-k MFMA window
                const int st = kt & 1;
                const int nwin = (p.K - (kt_begin_lo + kt) * BK8) > 64 ? 2 : 1;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    if (w < nwin) {
                        rd8(st, w);
                        if (w == 0 && kt + 1 < ntot) issue_tile(kt + 1, st ^ 1);
                        if (w == nwin - 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        bar1();
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af8[i], bf8[j], acc[i][j], 0, 0, 0, E8M0_LO_INV, 0, p.w_lo_exp);
                        bar2();
                    }
                }
            }
            for (int kt = n8; kt < ntot; ++kt) {
                const int st = kt & 1;
                const bool nxt = kt + 1 < ntot;
                static_for<4>([&](auto ph_) {
                    constexpr int ph = decltype(ph_)::value;
                    rd(st, ph);
                    if (nxt) {
                        // the next tile is a plain fp16 tile inside K: lane offsets fixed over the loop, the K tile as the scalar offset —
                        // no per-piece test on this path (the general issue_part() with its uniform branches on lo / K tail made the
                        // phase's load part longer than its MFMA part: measured 5-10 % slower than the un-staggered loop)
                        const unsigned ks = (unsigned)(kt_begin - nt_lo + kt + 1) * (BK * 2);
                        char* sa = smem + (st ^ 1) * STAGE + wave * 1024;
                        char* sb = sa + A_BYTES;
                        constexpr int QA = ph == 0 ? 0 : (ph == 1 ? Q0 : Q0 + Q1), QB = ph == 0 ? Q0 : (ph == 1 ? Q0 + Q1 : (ph == 2 ? LOADS : 0));
#pragma unroll
                        for (int q = QA; q < QB; ++q) {
                            if (q < A_IT) glds16_buf(rs_a, aoff[q < A_IT ? q : 0], ks, sa + q * (RPI * 128));
                            else glds16_buf(rs_w, woff[q >= A_IT ? q - A_IT : 0], ks, sb + (q - A_IT) * (RPI * 128));
                        }
                    }
                    if (ph == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    bar1();
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    bar2();
                });
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();               // group 0 waits for group 1's last phase
        }
    } else if (STAGES == 2) {
        // one tile in flight: the plain barrier carries the vmcnt(0) that lands the DMA
        issue_tile(0, 0);
        __syncthreads();
        // The second-dispatched half of the waves (4-7: one per SIMD, the arbitration losers) issues its share of the
        // next tile's DMA in the MIDDLE of its MFMA stream instead of together with waves 0-3 right after the barrier
        // (s_memtime timeline: 1870 vs 690 cycles per tile in the issue segment, with waves 0-3 then idling ~1400
        // cycles at the barrier): each SIMD then has one wave issuing DMA while the other runs MFMAs.
        const bool late = NW == 8 && wave >= 4 && wave_on && ntot >= 8;   // 2-5 % at long K
        // The e4m3 lo tiles run in a loop of their own (same pipeline, same tile counter): one MFMA kind per loop keeps the
        // register allocator from moving accumulator blocks between the two passes (a shared loop spilled 170 registers).
        const int n8 = lo8 ? nt_lo : 0;
        for (int kt = 0; kt < n8; ++kt) {              // (no mid-stream DMA issue here: the lo pass is 3-20 short tiles)
            if (kt + 1 < ntot) issue_tile(kt + 1, (kt + 1) & 1);
            if (wave_on) compute8(kt & 1, kt);
            __syncthreads();
        }
        for (int kt = n8; kt < ntot; ++kt) {
            const bool nxt = kt + 1 < ntot;
            if (nxt && !late) issue_tile(kt + 1, (kt + 1) & 1);
            if (wave_on) {
                compute(kt & 1, (nxt && late) ? kt + 1 : -1);
                if (!lo8 && kt + 1 == nt_lo) scale_lo();
            }
            __syncthreads();
        }
    } else {
        // ring of three stages, TWO tiles in flight.  Counted waits: after issuing tile kt+2 only its LOADS
        // DMA instructions may stay outstanding, i.e. tile kt+1 has landed; the raw s_barrier (no compiler
        // vmcnt(0)) then publishes every wave's part of it and retires all reads of the stage being recycled.
        issue_tile(0, 0);
        if (ntot > 1) {
            issue_tile(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        int st = 0;
        const int n8 = lo8 ? nt_lo : 0;
        for (int kt = 0; kt < n8; ++kt) {                             // e4m3 lo tiles (see the two-stage loop)
            const bool ahead = (kt + 2) < ntot;
            if (ahead) issue_tile(kt + 2, st == 0 ? 2 : st - 1);
            if (wave_on) compute8(st, kt);
            if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            st = (st == 2) ? 0 : st + 1;
        }
        for (int kt = n8; kt < ntot; ++kt) {
            const bool ahead = (kt + 2) < ntot;
            if (ahead) issue_tile(kt + 2, st == 0 ? 2 : st - 1);      // (kt + 2) % 3
            if (wave_on) {
                compute(st);
                if (!lo8 && kt + 1 == nt_lo) scale_lo();
            }
            if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            st = (st == 2) ? 0 : st + 1;
        }
    }

    // ------------------------------ epilogue ------------------------------
    if (!wave_on) return;                       // row group of another workgroup (tail split)
    const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
    if constexpr ((EPI & E_GENERIC) != 0) {
        epi_generic<MI, NI>(p, acc, lane, mw, nw);
    } else {
        if constexpr ((EPI & E_VT) != 0) {
            if (n0 >= p.n_split) { epi_vt<MI, NI>(p, acc, lane, mw, nw); return; }
        }
        if constexpr (EPI == (E_R1 | E_O32) || EPI == E_O32) {
            // (uniform) full tile, no activation: the direct epilogue (round 6; PNC_OPT_GEMM_FUSE_LN + 2, A/B: the staged one)
            if (direct_epi && p.act == PNC_ACT_NONE && m0 + BM <= p.M && n0 + BN <= p.N && !split_rows && ksplit == 1) {
                epi_direct_o32<MI, NI, (EPI & E_R1) != 0>(p, acc, lane, RowLinear{mw}, nw);
                return;
            }
        }
        float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EPITCH);
        __syncthreads();                        // every wave is done reading operand tiles from LDS
        if constexpr (GEGLU) {
            epi_geglu<MI, NI>(p, acc, ep, lane, mw, nw, reinterpret_cast<const float*>(smem + RING_BYTES));
        } else {
            if constexpr ((EPI & E_LN) != 0) {
                static_assert(WGN == 2, "the fused LayerNorm pairs the two waves of a row");
                float2* lnb = reinterpret_cast<float2*>(reinterpret_cast<float*>(smem) + NW * (32 * EPITCH));
                epi_fast<MI, NI, EPI>(p, acc, ep, lane, RowLinear{mw}, nw, p.N, lnb + wave * 64, lnb + (wave ^ 1) * 64);
            } else {
                float* gs_tab = reinterpret_cast<float*>(smem) + NW * (32 * EPITCH) + wave * (NI * 16);
                epi_fast<MI, NI, (EPI & ~E_VT)>(p, acc, ep, lane, RowLinear{mw}, nw, (EPI & E_VT) ? p.n_split : p.N, nullptr, nullptr,
                                                (EPI & E_GS) ? gs_tab : nullptr);   // E_GELU rides along
            }
        }
    }
}

// Workgroups of one geometry that are resident at once on the 256 CUs (LDS-limited: 160 KB per CU)
template <int LDS_BYTES>
constexpr int resident_slots() { return 256 * ((160 * 1024) / LDS_BYTES < 1 ? 1 : (160 * 1024) / LDS_BYTES > 2 ? 2 : (160 * 1024) / LDS_BYTES); }

// Tail split decision: tiles = q * slots + r.  When the last, partial round holds r <= slots/2 (or /4) tiles, run each of
// them as 2 (4) workgroups of BM/2 (BM/4) rows so that the round fills the chip: e.g. M = 49152, N = 640 with 256x320
// tiles is 384 tiles = 1.5 rounds -> 256 full tiles + 128 tiles x 2 halves.
template <int BM, int WGM, int LDS_BYTES>
static inline void tail_split(int tiles, int& nfull, int& tail_f) {
    constexpr int slots = resident_slots<LDS_BYTES>();
    const int r = tiles % slots;
    nfull = tiles; tail_f = 1;
    if (r == 0 || !pnc_get_option(PNC_OPT_GEMM_TAIL_SPLIT)) return;
    if (WGM >= 4 && r * 4 <= slots) tail_f = 4;
    else if (WGM >= 2 && r * 2 <= slots) tail_f = 2;
    if (tail_f > 1) nfull = tiles - r;
}

// gemm_stencil_tile.hip
int conv3x3_tile_geometry(const PncGemmParams& p, unsigned epi);
int dispatch_conv3x3_tiles(const PncGemmParams& p, unsigned epi, int geometry, hipStream_t st);

// gemm.hip
int launch_splitk_reduce(const PncGemmParams& p, int ksplit, hipStream_t st);

const float* phi_table_device(hipStream_t st, int* rc);

// PERSISTENT form of the GEGLU GEMM (FF1; PNC_OPT_GEMM_PERSIST, on by default) — the first step of DESIGN.md section 12a: one workgroup
// per CU walks its share of the output tiles, and the FIRST K tile of the next output tile is requested before the epilogue of the
// current one, into the ring stage the epilogue does not use (the register GEGLU epilogue stages 8 KB per wave = exactly one 64 KB
// stage).  Hidden: the ~2.5 us of DMA latency every tile otherwise starts with, the Phi table's reload and the workgroup turnover.
// Same tiles, same K order, same epilogue: bit-identical to the one-tile-per-workgroup kernel.  Preconditions (host): plain A, no lo
// plane, M % BM == 0, N % BN == 0, K % 64 == 0, fp16 output without lo plane.  Measured (profiles/round3/persist_ab_r3w.txt):
// level-0 FF1 472-540 -> 409-430 us, level 1 356 -> 334, level 2 316 -> 308.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_geglu_persist_kernel(const PncGemmParams pin, const float* __restrict__ phi_g,
                                                                            const int group_m, const int stagger_min_in) {
    PncGemmParams p = pin;
    constexpr int NW = WGM * WGN, MI = BM / WGM / 32, NI = BN / WGN / 32, RPI = NW * 8, A_IT = BM / RPI, B_IT = BN / RPI;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, RING_BYTES = 2 * STAGE;
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must be a multiple of the DMA row group");
    static_assert(NI % 2 == 0, "GEGLU pairs value / gate column blocks inside a wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];         // the operand ring: the ONLY memory LDS-DMA writes
    // Everything the epilogue reads lives in LDS objects of its own: hipcc puts s_waitcnt vmcnt(0) in front of any LDS access that
    // may alias an LDS-DMA in flight — with the staging inside the ring (as in gemm_glds_kernel) the epilogue would wait for the
    // prefetched K tile before its first table read.  160 KB = ring 128 + table 16 + one 2 KB slab of staging per wave 16.
    __shared__ __attribute__((aligned(16))) float s_phi[PHI_BYTES / 4];
    __shared__ __attribute__((aligned(16))) half_t s_stage[NW][32 * 32];
    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);
    const int tiles_n = p.N / BN, tiles_m = p.M / BM, ntile = tiles_m * tiles_n, nk = p.K / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    const int frow = lane & 31, fk = lane >> 5;

    // virtual block v -> output tile: the XCD-contiguous ranges and the grouped (tm, tn) order of gemm_glds_kernel
    auto tile_origin = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntile);
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
        m0 = tm * BM; n0 = tn * BN;
    };
    unsigned aoff[A_IT], woff[B_IT];                       // per-lane byte offsets inside a tile's windows: the same for every tile
#pragma unroll
    for (int i = 0; i < A_IT; ++i) aoff[i] = (unsigned)((i * RPI + srow) * p.lda + schunk * 8) * 2u;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) woff[i] = (unsigned)((i * RPI + srow) * p.ldw + schunk * 8) * 2u;
    auto issue_part = [&](int m0, int n0, int kt, int stage, auto q0_, auto q1_) __attribute__((always_inline)) {
        constexpr int Q0 = decltype(q0_)::value, Q1 = decltype(q1_)::value;       // DMA pieces [Q0, Q1): A row groups, then W row groups
        const buffer_rsrc_t rs_a = make_rsrc(A + (int64_t)m0 * p.lda, 0x7FFFFF00u);
        const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);
        char* sa = smem + stage * STAGE + wave * 1024;
        char* sb = sa + A_BYTES;
        const unsigned ks = (unsigned)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (i >= Q0 && i < Q1) glds16_buf(rs_a, aoff[i], ks, sa + i * (RPI * 128));
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (A_IT + i >= Q0 && A_IT + i < Q1) glds16_buf(rs_w, woff[i], ks, sb + i * (RPI * 128));
    };
    auto issue = [&](int m0, int n0, int kt, int stage) __attribute__((always_inline)) {
        issue_part(m0, n0, kt, stage, std::integral_constant<int, 0>{}, std::integral_constant<int, A_IT + B_IT>{});
    };

    for (int i = tid; i < PHI_BYTES / 16; i += 64 * NW)                 // the Phi table: once per workgroup, by plain stores
        reinterpret_cast<f32x4*>(s_phi)[i] = reinterpret_cast<const f32x4*>(phi_g)[i];
    f32x16 acc[MI][NI];
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        half8v af[2][MI], bf[2][NI];
        auto frags = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[b][i] = *reinterpret_cast<const half8v*>(sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bf[b][j] = *reinterpret_cast<const half8v*>(sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks + 1 < BK / 16) frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    const bool direct16 = (stagger_min_in & 256) == 0;     // (+ 256, A/B: the round-3 epilogue through a 2 KB LDS slab per wave)
    const int stagger_min = stagger_min_in & 255;
    if (NW == 8 && wave >= 4 && !(stagger_min > 0 && nk >= stagger_min)) __builtin_amdgcn_s_setprio(1);
    int v = blockIdx.x;
    if (v >= ntile) return;
    int m0, n0, sp = 0;
    tile_origin(v, m0, n0);
    float pb[NI], pbn[NI];
    auto load_bias = [&](int n0_, float (&dst)[NI]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) dst[j] = p.bias ? p.bias[n0_ + wn * (NI * 32) + j * 32 + (lane & 31)] : 0.0f;
    };
    load_bias(n0, pb);
    issue(m0, n0, 0, 0);
    // staggered schedule of the K loop (gemm_glds_kernel; PNC_OPT_GEMM_STAGGER): waves 4-7 one barrier behind waves 0-3 inside an
    // output tile's K loop, both groups aligned again before the epilogue (their epilogues run together, as before; run one behind
    // the other they would serialise: a group can do ONE phase while the other is in its epilogue).  The next output tile's first K
    // tile is requested in phases 0-2 of the LAST K tile instead of in front of the epilogue.
    const bool staggered = NW == 8 && stagger_min > 0 && nk >= (stagger_min == 1 ? 1 : stagger_min);
    const int grp = wave >> 2;
    while (true) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        const int ls = (sp + nk - 1) & 1;       // the stage of the last K tile: every wave is done with it -> the epilogue's staging
        const int vn = v + gridDim.x;
        int m1 = 0, n1 = 0;
        if (staggered) {
            constexpr int LOADS = A_IT + B_IT, Q0 = (LOADS + 2) / 3, Q1 = (LOADS - Q0 + 1) / 2;
            const std::integral_constant<int, 0> C0{};
            const std::integral_constant<int, Q0> CQ0{};
            const std::integral_constant<int, Q0 + Q1> CQ1{};
            const std::integral_constant<int, LOADS> CQ2{};
            if (v == (int)blockIdx.x) {          // (uniform) first output tile: its K tile 0 was requested above (+ the Phi table's stores)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }                                    // (later tiles: landed and published by the previous tile's last phase)
            if (grp == 1) __builtin_amdgcn_s_barrier();
            if (vn < ntile) tile_origin(vn, m1, n1);
            half8v af[MI], bf[NI];
            for (int kt = 0; kt < nk; ++kt) {
                const int st = (sp + kt) & 1;
                const bool last = kt + 1 == nk;
                const bool nxt = !last || vn < ntile;
                const int nm0 = last ? m1 : m0, nn0 = last ? n1 : n0, nkt = last ? 0 : kt + 1;
#pragma unroll
                for (int ph = 0; ph < 4; ++ph) {
                    const char* sa = smem + st * STAGE;
                    const char* sb = sa + A_BYTES;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        af[i] = *reinterpret_cast<const half8v*>(sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ph * 2 + fk));
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        bf[j] = *reinterpret_cast<const half8v*>(sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ph * 2 + fk));
                    if (nxt) {
                        if (ph == 0) {
                            if (last) load_bias(n1, pbn);         // BEFORE the DMA (vmcnt is in order)
                            issue_part(nm0, nn0, nkt, st ^ 1, C0, CQ0);
                        } else if (ph == 1) issue_part(nm0, nn0, nkt, st ^ 1, CQ0, CQ1);
                        else if (ph == 2) issue_part(nm0, nn0, nkt, st ^ 1, CQ1, CQ2);
                    }
                    if (ph == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                }
            }
            if (grp == 0) __builtin_amdgcn_s_barrier();          // both groups past their last phase: the epilogues start together
        } else {
        __syncthreads();                        // K tile 0 of this output tile has landed; the previous epilogue's staging is retired
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(m0, n0, kt + 1, (sp + kt + 1) & 1);
            compute((sp + kt) & 1);
            __syncthreads();
        }
        if (vn < ntile) {                       // (uniform) the next output tile's first K tile, into the other stage
            tile_origin(vn, m1, n1);
            load_bias(n1, pbn);                 // BEFORE the DMA: nothing in the epilogue below may wait on vmcnt
            issue(m1, n1, 0, ls ^ 1);
        }
        }
        {   // epi_geglu's register path, slab by slab (same operations in the same order: bit-identical)
            half_t* out16 = reinterpret_cast<half_t*>(p.out16);
            typedef half_t __attribute__((may_alias)) half_st;
            typedef int4 __attribute__((may_alias)) int4_st;
            half_st* sb = reinterpret_cast<half_st*>(&s_stage[wave][0]);
            const int c = lane & 31, cl = lane & 3, rl = lane >> 2;
            const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
            static_for<NI / 2>([&](auto jc_) {
                constexpr int jc = decltype(jc_)::value * 2;
                const float bv = pb[jc], bg = pb[jc + 1];
                const int ncol0 = (nw + jc * 32) >> 1;
                static_for<MI>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    float gx[16], fr[16];
                    int ix[16];
                    float2 e[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        gx[r] = acc[i][jc + 1][r] + bg;
                        float t = fmaf(gx[r], PHI_SCALE, -PHI_X0 * PHI_SCALE);
                        t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)PHI_N - 0.001f);
                        ix[r] = (int)t;
                        fr[r] = t - (float)ix[r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = *reinterpret_cast<const float2*>(s_phi + 2 * ix[r]);
                    if (direct16) {
                        // round 6: the products leave through two-byte buffer stores straight from the registers (a lane holds one column of
                        // rows 8 q + 4 h + e: two 64-byte row pieces per instruction, the row inside the block as the scalar offset) instead of
                        // through the slab (16 two-byte staging writes + 2 reads + 2 sixteen-byte stores per
                        // block): FF1 −0.5 … −1.7 % at every level, step −0.3 ms (profiles/round6/ff1_direct_stores_r6.log).  Same values.
                        const buffer_rsrc_t ro = make_rsrc(out16 + (int64_t)(mw + i * 32) * p.ldc16, 0x7FFFFF00u);
                        const int vo = (4 * (lane >> 5) * p.ldc16 + ncol0 + c) * 2;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float prod = (acc[i][jc][r] + bv) * (gx[r] * fmaf(fr[r], e[r].y, e[r].x));
                            asm("" : "+v"(prod));
                            const half_t hp = (half_t)prod;
                            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hp), ro, vo, ((r & 3) + 8 * (r >> 2)) * p.ldc16 * 2, 0);
                        }
                    } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float prod = (acc[i][jc][r] + bv) * (gx[r] * fmaf(fr[r], e[r].y, e[r].x));
                        asm("" : "+v"(prod));
                        sb[mfma32_row(r, lane) * 32 + c] = (half_t)prod;
                    }
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = ps * 16 + rl;
                        const int4 v4 = *reinterpret_cast<const int4_st*>(sb + row * 32 + cl * 8);
                        *reinterpret_cast<int4_st*>(out16 + (int64_t)(mw + i * 32 + row) * p.ldc16 + ncol0 + cl * 8) = v4;
                    }
                    }
                });
            });
        }
        if (vn >= ntile) break;
        v = vn; m0 = m1; n0 = n1; sp = ls ^ 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) pb[j] = pbn[j];
    }
}

// (Round 4 measured the same GEMM as TWO independent persistent 4-wave workgroups per CU — 128 x 256 tiles, 32-channel half tiles,
// bit-identical — so that one workgroup's GEGLU epilogue runs under the other's MFMAs, the arrangement that gave the view
// attention 8 %: 6 % SLOWER at level 0 (410 -> 435 us), 14 % at levels 1-2; staging W once per 128 rows instead of once per 256
// costs more than the overlap returns.  Kernel source and numbers: tools/exp/gemm_geglu_2wg_kernel.h, profiles/round4/ff1_two_workgroups_ab_r4i.txt.)
template <int BM, int BN, int WGM, int WGN>
int launch_geglu_persist(const PncGemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * 128;               // dynamic part: the operand ring (table and staging are static)
    static_assert(lds + PHI_BYTES + WGM * WGN * 2048 <= 160 * 1024, "LDS budget of one CU");
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = gemm_geglu_persist_kernel<BM, BN, WGM, WGN>;
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    int rc = PNC_OK;
    const float* phi = phi_table_device(st, &rc);
    if (rc != PNC_OK) return rc;
    const int tiles_n = p.N / BN, tiles_m = p.M / BM, tiles = tiles_m * tiles_n;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? gopt : (tiles_n > 8 ? 4 : 0);
    if (group_m > tiles_m) group_m = tiles_m;
    if (tiles_n < 2) group_m = 0;
    static std::atomic<int> ncu_of[64];                    // CUs per device, asked once (one persistent workgroup per CU)
    int ncu = ncu_of[dev & 63].load(std::memory_order_relaxed);
    if (ncu == 0) {
        int v = 0;
        ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        ncu_of[dev & 63].store(ncu, std::memory_order_relaxed);
    }
    const int blocks = tiles < ncu ? tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WGM * WGN), lds, st, p, phi, group_m, pnc_get_option(PNC_OPT_GEMM_STAGGER));
    return pnc_launch_status();
}
// the persistent kernel serves this problem (and the switch is on)
static inline bool geglu_persist_ok(const PncGemmParams& p) {
    return (pnc_get_option(PNC_OPT_GEMM_PERSIST) & 1) != 0 && p.geglu && p.a_mode == PNC_A_PLAIN && !p.A_lo && !p.out16_lo && p.out16 &&
           (p.M % 256) == 0 && (p.N % 256) == 0 && (p.K % 64) == 0 && p.K >= 64 && (p.M / 256) * (p.N / 256) >= 512;
}

// ---------------------------------------------------------------------------------------------------------------
// PERSISTENT form of the plain-A 256x320 GEMM for the fp32-epilogue families (round 4; PNC_OPT_GEMM_PERSIST bit 1): C x C
// projections with the residual stream in place (+ fused LayerNorm), proj_in (+ position table) + norm1, FF2 — the launches
// that are 0.09-0.25 of the MFMA peak because a 5-10 K-tile main loop and a 630 MB epilogue stream simply ADD
// (profiles/round3/fixed_cost_per_tile_r3q.txt).  One workgroup per CU walks its share of the output tiles and requests the
// FIRST K tile of the next output tile before the epilogue of the current one, as gemm_geglu_persist_kernel does for FF1.
// What FF1 did not need: these epilogues stage fp32 slabs of 8.7 KB per wave INSIDE the operand ring (2 x 72 KB of the 160 KB;
// no room for a staging array of its own), and hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS access it can see that may
// alias an LDS-DMA in flight — which would drain the prefetch AND every residual load / store of the rolling pipeline (vmcnt
// counts stores on gfx950).  So the epilogue's LDS traffic is written as inline asm (ds_write / ds_read + counted lgkmcnt
// waits that carry the registers they guard): the staging lives in stage 0 (exactly its 73 728 bytes incl. the LayerNorm
// row sums), the prefetch goes to stage 1, and the next tile's K loop starts there.  Same tiles, K order, MFMA order and
// epilogue arithmetic as gemm_glds_kernel<PNC_A_PLAIN, 256, 320, 4, 2, 2, false, EPI>: bit-identical
// (tests/test_kernels_gpu.py::test_gemm_persistent_kernel_is_bit_identical).
__device__ __forceinline__ void alds_w32(unsigned a, float v, int off) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(off) : "memory");
}
template <int OFF>
__device__ __forceinline__ void alds_w32c(unsigned a, float v) { asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ float alds_r32c(unsigned a) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void alds_w128c(unsigned a, f32x4 v) { asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ f32x4 alds_r128c(unsigned a) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int OFF = 0>
__device__ __forceinline__ float2 alds_r64(unsigned a) {
    float2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}
template <int OFF = 0>
__device__ __forceinline__ void alds_w64(unsigned a, float2 v) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory"); }
// wait until at most N of this wave's LDS operations are outstanding; the registers the wait guards are operands, so that no
// use of them can be scheduled above it
template <int N>
__device__ __forceinline__ void alds_wait(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
__device__ __forceinline__ void alds_wait0(float2& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void alds_wait0(float2& a, float2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory"); }
__device__ __forceinline__ void alds_wait0(f32x16& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void alds_wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// epi_fast() of a full wave tile (every row / column inside the matrix: the host guarantees M % BM == 0, N % BN == 0) with all
// LDS traffic in asm.  `ep`: LDS byte address of this wave's staging region (32 x EPITCH floats), `ln_mine` / `ln_partner`:
// byte addresses of the two 64-row float2 statistics arrays of the wave pair that shares the rows.  One added stream at most
// (res1 or the row bias).  The arithmetic — order of the additions, one-pass LayerNorm statistics — is epi_fast's.
template <int MI, int NI, unsigned EPI>
__device__ __forceinline__ void epi_fast_alds(const PncGemmParams& p, f32x16 (&acc)[MI][NI], unsigned ep, int lane_in,
                                              int mw, int nw, unsigned ln_mine, unsigned ln_partner) {
    // every lane constant of the epilogue hangs off this opaque copy: computed per call (= per output tile), not hoisted out of
    // the persistent kernel's tile loop into ~40 more registers that live through the main loop
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    constexpr bool R1 = (EPI & E_R1) != 0, RB = (EPI & E_RB) != 0;
    constexpr bool O32 = (EPI & E_O32) != 0, O16 = (EPI & E_O16) != 0, LN = (EPI & E_LN) != 0;
    static_assert((EPI & (E_R2 | E_VT | E_GEGLU | E_GELU | E_GENERIC)) == 0, "persistent variants: one added stream, row-major outputs");
    static_assert(!(R1 && RB), "one added stream");
    static_assert(!LN || O32, "the fused LayerNorm normalises the fp32 output it has just written");
    constexpr bool HAS_X = R1 || RB;
    constexpr int ENI = 2, EPITCH = ENI * 32 + 4;
    constexpr int NJ = (NI + ENI - 1) / ENI, NS = NJ * MI, NPMAX = 4;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    void* out16_lo = p.out16_lo;
    const bool silu = (!HAS_X) && (p.act == PNC_ACT_SILU);
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 x0[NPMAX], x1[NPMAX];
    // E_LN: the final fp32 values of the wave tile stay in registers between the two passes, in the ROW-MAJOR layout the passes
    // work in (8 consecutive columns of one row per lane and pass) — they take the place of the accumulator blocks, which are dead
    // once staged.  (epi_fast writes them back to the slab and reloads the MFMA layout, then stages a second time: 72 more LDS
    // instructions per slab, and the reload keeps all 160 accumulator registers live through pass 1.)
    float fin[LN ? NS : 1][NPMAX][8];
    if constexpr (LN) {
        if (lane < MI * 32) alds_w64(ln_mine + (unsigned)lane * 8u, make_float2(0.0f, 0.0f));
    }
    // staging addresses: a lane WRITES accumulator register r of column block j at row mfma32_row(r, lane), column j*32 + (lane & 31)
    // and READS 8 consecutive columns cl*8 of row ps*RPP + rl; everything but the lane part is an instruction offset
    const unsigned wbase = ep + (unsigned)((4 * (lane >> 5)) * EPITCH + (lane & 31)) * 4u;

    auto load_x = [&](auto s_, auto ps_) {
        constexpr int s = decltype(s_)::value, ps = decltype(ps_)::value;
        constexpr int jc = (s / MI) * ENI, i = s % MI, cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        constexpr int CPL = cw * 4, RPP = 64 / CPL;
        const int cl = lane % CPL, rl = lane / CPL;
        const int ncol = nw + jc * 32 + cl * 8;
        const int m = mw + i * 32 + ps * RPP + rl;
        const float* xp = R1 ? p.res1 + (int64_t)m * p.ldr1 + ncol
                             : p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
        x0[ps] = ld4(xp); x1[ps] = ld4(xp + 4);
    };

    static_for<NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value, jc = (s / MI) * ENI, i = s % MI;
        constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        constexpr int CPL = cw * 4, RPP = 64 / CPL, NP = 32 / RPP;
        const int cl = lane % CPL, rl = lane / CPL;
        const int ncol = nw + jc * 32 + cl * 8;
        const unsigned rbase = ep + (unsigned)(rl * EPITCH + cl * 8) * 4u;
        f32x4 b0 = z4, b1 = z4;
        if (p.bias) { b0 = ld4(p.bias + ncol); b1 = ld4(p.bias + ncol + 4); }
        static_for<cw>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<16>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                alds_w32c<(((r & 3) + 8 * (r >> 2)) * EPITCH + j * 32) * 4>(wbase, acc[i][jc + j][r]);
            });
        });
        if constexpr (s == 0 && HAS_X) {
            static_for<NP>([&](auto ps_) { load_x(std::integral_constant<int, 0>{}, ps_); });
        }
        // (a wave's LDS operations execute in order: the reads below see the writes above without a wait in between)
        // The passes run in batches of NB: their staged rows are read together, consumed under counted waits.  (All NP passes
        // at once hold 32 registers of staged values next to the 32 of the rolling stream prefetch and the 160 accumulators:
        // the LayerNorm variants then spilled 40-80 VGPRs.)
        constexpr int NB = (NP < 2 || (LN && HAS_X)) ? 1 : 2;     // (LayerNorm + added stream: one pass at a time, no spill)
        static_for<NP / NB>([&](auto bt_) {
        constexpr int bt = decltype(bt_)::value;
        f32x4 a0[NB], a1[NB];
        static_for<NB>([&](auto q_) {
            constexpr int q = decltype(q_)::value, ps = bt * NB + q;
            a0[q] = alds_r128c<ps * RPP * EPITCH * 4>(rbase);
            a1[q] = alds_r128c<ps * RPP * EPITCH * 4 + 16>(rbase);
        });
        static_for<NB>([&](auto q_) {
            constexpr int q = decltype(q_)::value, ps = bt * NB + q;
            std::integral_constant<int, ps> ps_;
            alds_wait<2 * (NB - 1 - q)>(a0[q], a1[q]);
            const int m = mw + i * 32 + ps * RPP + rl;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[q][e] + b0[e]; v[e + 4] = a1[q][e] + b1[e]; }
            if constexpr (HAS_X) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += x0[ps][e]; v[e + 4] += x1[ps][e]; }
            } else {
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                }
            }
            if constexpr (HAS_X && s + 1 < NS) {
                constexpr int jn = ((s + 1) / MI) * ENI, cwn = (NI - jn) < ENI ? (NI - jn) : ENI;
                constexpr int NPn = 32 / (64 / (cwn * 4));
                if constexpr (ps < NPn) load_x(std::integral_constant<int, s + 1>{}, ps_);
            }
            if constexpr (LN) {
                float sm = 0.0f, sq = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { sm += v[e]; sq = fmaf(v[e], v[e], sq); }
#pragma unroll
                for (int o = 1; o < CPL; o <<= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }
                if (cl == 0) {
                    const unsigned la = ln_mine + (unsigned)rl * 8u;           // + (i * 32 + ps * RPP) rows as the instruction offset
                    float2 t = alds_r64<(i * 32 + ps * RPP) * 8>(la);
                    alds_wait0(t);
                    t.x += sm; t.y += sq;
                    alds_w64<(i * 32 + ps * RPP) * 8>(la, t);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) fin[s][ps][e] = v[e];
            }
            if constexpr (O32) {
                float* op = p.out32 + (int64_t)m * p.ldc32 + ncol;
                *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
            if constexpr (O16) store_h8(out16, out16_lo, (int64_t)m * p.ldc16 + ncol, v, p.out_lo_fmt);
        });
        });
    });
    if constexpr (LN) {
        alds_wait0();
        __builtin_amdgcn_s_barrier();             // both halves of every row have their statistics in LDS
        half_t* lnout = reinterpret_cast<half_t*>(p.ln_out16);
        const float invn = 1.0f / (float)p.N;
        static_for<NS>([&](auto s_) {
            constexpr int s = decltype(s_)::value, jc = (s / MI) * ENI, i = s % MI;
            constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;
            constexpr int CPL = cw * 4, RPP = 64 / CPL, NP = 32 / RPP;
            const int cl = lane % CPL, rl = lane / CPL;
            const int ncol = nw + jc * 32 + cl * 8;
            const f32x4 g0 = ld4(p.ln_gamma + ncol), g1 = ld4(p.ln_gamma + ncol + 4);
            const f32x4 h0 = ld4(p.ln_beta + ncol), h1 = ld4(p.ln_beta + ncol + 4);
            constexpr int NB = NP < 2 ? NP : 2;
            static_for<NP / NB>([&](auto bt_) {
            constexpr int bt = decltype(bt_)::value;
            float2 sa[NB], sb[NB];
            static_for<NB>([&](auto q_) {
                constexpr int q = decltype(q_)::value, ps = bt * NB + q;
                sa[q] = alds_r64<(i * 32 + ps * RPP) * 8>(ln_mine + (unsigned)rl * 8u);
                sb[q] = alds_r64<(i * 32 + ps * RPP) * 8>(ln_partner + (unsigned)rl * 8u);
            });
            static_for<NB>([&](auto q_) {
                constexpr int q = decltype(q_)::value, ps = bt * NB + q;
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(sa[q]), "+v"(sb[q]) : "n"(2 * (NB - 1 - q)) : "memory");
                const int m = mw + i * 32 + ps * RPP + rl;
                const float mean = (sa[q].x + sb[q].x) * invn;
                const float var = fmaxf(fmaf(-mean, mean, (sa[q].y + sb[q].y) * invn), 0.0f);     // = epi_fast's, written out
                const float rs = rsqrtf(var + p.ln_eps);
                float y[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = fmaf((fin[s][ps][e] - mean) * rs, g0[e], h0[e]);
                    y[e + 4] = fmaf((fin[s][ps][e + 4] - mean) * rs, g1[e], h1[e]);
                }
                store_h8(lnout, nullptr, (int64_t)m * p.ldln + ncol, y);
            });
            });
        });
    }
}

template <unsigned EPI>
__global__ __launch_bounds__(512) void gemm_persist_kernel(const PncGemmParams pin, const int group_m_in) {
    PncGemmParams p = pin;
    const int group_m = group_m_in & 0xFFFF;
    constexpr int BM = 256, BN = 320, WGM = 4, WGN = 2;
    constexpr int NW = WGM * WGN, MI = BM / WGM / 32, NI = BN / WGN / 32, RPI = NW * 8, A_IT = BM / RPI, B_IT = BN / RPI;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int EPITCH = 2 * 32 + 4;
    static_assert(NW * (32 * EPITCH * 4 + 64 * 8) <= STAGE, "the epilogue's staging (+ LayerNorm row sums) lives in stage 0");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);
    const int tiles_n = p.N / BN, tiles_m = p.M / BM, ntile = tiles_m * tiles_n;
    const bool lo8 = p.A_lo != nullptr;                       // (host: e4m3 lo planes only)
    const int ntiles = p.K / BK;
    const int nt_lo = lo8 ? (p.K + BK8 - 1) / BK8 : 0;
    const int ntot = ntiles + nt_lo;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    auto tile_origin = [&](int v, int& m0, int& n0) {         // the XCD-contiguous ranges and the grouped order of gemm_glds_kernel
        const int tile = xcd_remap(v, ntile);
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
        m0 = tm * BM; n0 = tn * BN;
    };
    f32x16 acc[MI][NI];
    // the lane constants of the operand DMA and of the fragment reads (row / chunk split, per-lane window offsets, LDS row
    // addresses: ~25 VGPRs) are derived from an opaque copy of the lane id INSIDE the tile loop: hoisted out of it they stay live
    // through the epilogue, next to 160 accumulator registers and the epilogue's own 64, and the LayerNorm variants spilled 80
    auto lane_consts = [&](int& srow, int& schunk, int& frow, int& fk) {
        int lt = lane;
        asm volatile("" : "+v"(lt));
        srow = wave * 8 + (lt >> 3);
        schunk = (lt & 7) ^ ((srow >> 1) & 7);
        frow = lt & 31; fk = lt >> 5;
    };
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    int v = blockIdx.x;
    if (v >= ntile) return;
    int m0, n0, sp = 0;
    tile_origin(v, m0, n0);
    bool first = true;
    const unsigned smem_lds = (unsigned)(uintptr_t)smem;
    const unsigned ep = smem_lds + (unsigned)wave * (32 * EPITCH * 4);
    const unsigned lnb = smem_lds + NW * (32 * EPITCH * 4);
    const bool late = wave >= 4 && ntot >= 8;
    while (true) {
        int srow, schunk, frow, fk;
        lane_consts(srow, schunk, frow, fk);
        unsigned aoff[A_IT], woff[B_IT];                          // per-lane byte offsets inside a tile's windows: the same for every tile
#pragma unroll
        for (int i = 0; i < A_IT; ++i) aoff[i] = (unsigned)((i * RPI + srow) * p.lda + schunk * 8) * 2u;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) woff[i] = (unsigned)((i * RPI + srow) * p.ldw + schunk * 8) * 2u;
        // K tile kt_local of the output tile at (m0, n0) into `stage`: e4m3 lo tiles first (128 k per 128-byte row), then the fp16 tiles
        auto issue = [&](int m0, int n0, int kt_local, int stage) {
            char* sa = smem + stage * STAGE + wave * 1024;
            char* sb = sa + A_BYTES;
            if (kt_local < nt_lo) {
                int schunk8 = schunk, srow8 = srow;
                asm volatile("" : "+v"(schunk8), "+v"(srow8));
                const buffer_rsrc_t rs_alo = make_rsrc(reinterpret_cast<const char*>(p.A_lo) + (int64_t)m0 * p.lda, 0x7FFFFF00u);
                const buffer_rsrc_t rs_wlo = make_rsrc(reinterpret_cast<const char*>(p.W_lo) + (int64_t)n0 * p.ldw_lo, 0x7FFFFF00u);
                const int kc8 = kt_local * BK8 + schunk8 * 16;
                const unsigned ks8 = (unsigned)kt_local * BK8;
                // chunks of the last tile beyond K: bit 31 of the lane offset puts them past the resource's bound (zeros) — as an
                // OR, not a select: hipcc turns the select into two DMA instructions under complementary EXEC masks per piece
                const unsigned oob = (unsigned)(p.K - 1 - kc8) & 0x80000000u;
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    glds16_buf(rs_alo, ((aoff[i] >> 1) + (unsigned)schunk8 * 8u) | oob, ks8, sa + i * (RPI * 128));
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    glds16_buf(rs_wlo, (unsigned)((i * RPI + srow8) * p.ldw_lo + schunk8 * 16) | oob, ks8, sb + i * (RPI * 128));
                return;
            }
            const buffer_rsrc_t rs_a = make_rsrc(A + (int64_t)m0 * p.lda, 0x7FFFFF00u);
            const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);
            const unsigned ks = (unsigned)(kt_local - nt_lo) * (BK * 2);
#pragma unroll
            for (int i = 0; i < A_IT; ++i) glds16_buf(rs_a, aoff[i], ks, sa + i * (RPI * 128));
#pragma unroll
            for (int i = 0; i < B_IT; ++i) glds16_buf(rs_w, woff[i], ks, sb + i * (RPI * 128));
        };

        // fp16 K tile: the non-pipelined fragment order of the 256x320 geometry (gemm_glds_kernel, PIPE = false)
        auto compute = [&](int stage, int m0, int n0, int mid_kt, int mid_stage) {
            const char* sa = smem + stage * STAGE;
            const char* sb = sa + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                half8v af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    af[i] = *reinterpret_cast<const half8v*>(sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    bf[j] = *reinterpret_cast<const half8v*>(sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (ks == 1 && mid_kt >= 0) issue(m0, n0, mid_kt, mid_stage);
            }
        };
        auto compute8 = [&](int stage, int kt_local) {
            const char* sa = smem + stage * STAGE;
            const char* sb = sa + A_BYTES;
            const int nwin = (p.K - kt_local * BK8) > 64 ? 2 : 1;
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                if (w < nwin) {
                    i32x8 af[MI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int row = wm * (MI * 32) + i * 32 + frow;
                        const i32x4 a0 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2));
                        const i32x4 a1 = *reinterpret_cast<const i32x4*>(sa + lds_off128(row, w * 4 + fk * 2 + 1));
                        af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int row = wn * (NI * 32) + j * 32 + frow;
                        const i32x4 b0 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2));
                        const i32x4 b1 = *reinterpret_cast<const i32x4*>(sb + lds_off128(row, w * 4 + fk * 2 + 1));
                        const i32x8 bf = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf, acc[i][j], 0, 0, 0, E8M0_LO_INV, 0, p.w_lo_exp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };

#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        if (first) { issue(m0, n0, 0, 0); first = false; }      // (uniform) every later tile's first K tile was prefetched
        __syncthreads();                        // K tile 0 of this output tile has landed; the previous epilogue's staging is retired
        for (int kt = 0; kt < nt_lo; ++kt) {
            if (kt + 1 < ntot) issue(m0, n0, kt + 1, (sp + kt + 1) & 1);
            compute8((sp + kt) & 1, kt);
            __syncthreads();
        }
        for (int kt = nt_lo; kt < ntot; ++kt) {
            const bool nxt = kt + 1 < ntot;
            if (nxt && !late) issue(m0, n0, kt + 1, (sp + kt + 1) & 1);
            compute((sp + kt) & 1, m0, n0, (nxt && late) ? kt + 1 : -1, (sp + kt + 1) & 1);
            __syncthreads();
        }
        // every wave is past the last K tile: the ring is free.  The next output tile's first K tile goes to stage 1; the
        // epilogue stages in stage 0 through asm LDS operations, which the compiler does not order against the DMA
        const int vn = v + gridDim.x;
        int m1 = 0, n1 = 0;
        if (vn < ntile) {
            tile_origin(vn, m1, n1);
            issue(m1, n1, 0, 1);
        }
        if constexpr ((EPI & E_VT) != 0) {
            // column tiles at or beyond n_split go channel-major straight from the accumulators (no LDS at all)
            if (n0 >= p.n_split) epi_vt<MI, NI>(p, acc, lane, m0 + wm * (MI * 32), n0 + wn * (NI * 32));
            else epi_fast_alds<MI, NI, (EPI & ~E_VT)>(p, acc, ep, lane, m0 + wm * (MI * 32), n0 + wn * (NI * 32), 0u, 0u);
        } else if constexpr (EPI == E_O32 || EPI == (E_R1 | E_O32)) {
            // fp32 output (+ residual in place) straight from the accumulators (round 6, epi_direct_o32; group_m bit 16, A/B: staged)
            if (!(group_m_in & 0x10000) && p.act == PNC_ACT_NONE)
                epi_direct_o32<MI, NI, (EPI & E_R1) != 0>(p, acc, lane, RowLinear{m0 + wm * (MI * 32)}, n0 + wn * (NI * 32));
            else
                epi_fast_alds<MI, NI, EPI>(p, acc, ep, lane, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lnb + (unsigned)wave * 512u,
                                           lnb + (unsigned)(wave ^ 1) * 512u);
        } else {
            epi_fast_alds<MI, NI, EPI>(p, acc, ep, lane, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lnb + (unsigned)wave * 512u,
                                       lnb + (unsigned)(wave ^ 1) * 512u);
        }
        if (vn >= ntile) break;
        alds_wait0();                           // this wave's staging reads have returned before it meets the others at the barrier
        v = vn; m0 = m1; n0 = n1; sp = 1;
    }
}

// the persistent plain-A kernel serves this problem (PNC_OPT_GEMM_PERSIST bit 1)
static inline bool plain_persist_ok(const PncGemmParams& p, unsigned epi) {
    if ((pnc_get_option(PNC_OPT_GEMM_PERSIST) & 2) == 0 || p.a_mode != PNC_A_PLAIN) return false;
    if ((p.M % 256) || (p.N % 320) || (p.K % 64) || p.K < 64) return false;
    if (p.A_lo && (p.a_lo_fmt != PNC_LO_E4M3 || (p.lda % 16))) return false;
    if ((epi & E_LN) && p.N != 320) return false;
    if ((epi & E_VT) && ((p.n_split % 320) || (p.M % 8) || (p.t_rows % 8))) return false;
    return (p.M / 256) * (p.N / 320) >= 512;                 // at least two output tiles per workgroup
}

template <unsigned EPI>
int launch_plain_persist(const PncGemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * (256 + 320) * 128;
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = gemm_persist_kernel<EPI>;
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    const int tiles_n = p.N / 320, tiles_m = p.M / 256, tiles = tiles_m * tiles_n;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? gopt : (tiles_n > 8 ? 4 : 0);
    if (group_m > tiles_m) group_m = tiles_m;
    if (tiles_n < 2) group_m = 0;
    static std::atomic<int> ncu_of[64];
    int ncu = ncu_of[dev & 63].load(std::memory_order_relaxed);
    if (ncu == 0) {
        int v = 0;
        ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        ncu_of[dev & 63].store(ncu, std::memory_order_relaxed);
    }
    const int blocks = tiles < ncu ? tiles : ncu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, p, group_m | ((pnc_get_option(PNC_OPT_GEMM_FUSE_LN) & 2) ? 0x10000 : 0));
    return pnc_launch_status();
}

template <int AMODE, int BM, int BN, int WGM, int WGN, int STAGES, bool PIPE, unsigned EPI>
int launch(const PncGemmParams& p, hipStream_t st, int ksplit = 1) {
    constexpr int lds = STAGES * (BM + BN) * 128;
    constexpr int threads = 64 * WGM * WGN;
    constexpr bool GEGLU = (EPI & E_GEGLU) != 0;
    static_assert(lds + (GEGLU ? PHI_BYTES : 0) <= 160 * 1024, "LDS budget of one CU (operand ring + GEGLU table)");
    static_assert(lds >= WGM * WGN * (32 * 68 * 4 + 64 * 8), "epilogue staging (+ LayerNorm row sums) must fit the operand ring");
    static std::atomic<unsigned char> attr_done[64];      // per instantiation and device; the call is idempotent
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = gemm_glds_kernel<AMODE, BM, BN, WGM, WGN, STAGES, PIPE, EPI>;
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds + (GEGLU ? PHI_BYTES : 0));
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    int nfull = tiles, tail_f = 1;
    if (ksplit == 1) tail_split<BM, WGM, lds>(tiles, nfull, tail_f);
    const int blocks = ksplit > 1 ? tiles * ksplit : nfull + (tiles - nfull) * tail_f;
    PncGemmParams q = p;
    const float* phi = nullptr;
    if constexpr (GEGLU) {
        int rc = PNC_OK;
        phi = phi_table_device(st, &rc);
        if (rc != PNC_OK) return rc;
    }
    if (ksplit > 1) {            // raw partial sums; the reduce launch owns bias / residuals / outputs
        q.ldc32 = p.N;
        q.bias = nullptr; q.rowbias = nullptr; q.res1 = nullptr; q.res2 = nullptr;
        q.out16 = nullptr; q.out16_lo = nullptr; q.out16t = nullptr; q.n_split = p.N; q.act = PNC_ACT_NONE;
    }
    // grouped tile order (see the kernel): auto = 4 row panels per group when the problem has more than 8 column tiles
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? gopt : (tiles_n > 8 ? 4 : 0);           // 1 = plain order
    if (group_m > tiles_m) group_m = tiles_m;
    if (tiles_n < 2) group_m = 0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds + (GEGLU ? PHI_BYTES : 0), st, q, ksplit, nfull, tail_f, phi,
                       group_m, (pnc_get_option(PNC_OPT_GEMM_STAGGER) & 0xFF) | ((pnc_get_option(PNC_OPT_GEMM_FUSE_LN) & 2) ? 0 : 0x100));
    if (ksplit > 1) return launch_splitk_reduce(p, ksplit, st);
    return pnc_launch_status();
}

// expected relative throughput of a geometry on `slots` concurrently resident workgroups
static inline double tile_score(long tiles, int slots, double eff) {
    if (tiles <= 0) return 0.0;
    // a partial last round costs a full round, unless tail_split() can run it as half / quarter tiles (the row-split
    // workgroups still stage the whole W tile: ~0.65 / 0.45 of a full tile's time)
    const long q = tiles / slots, r = tiles % slots;
    const double tail = r == 0 ? 0.0 : (r * 4 <= slots ? 0.45 : (r * 2 <= slots ? 0.65 : 1.0));
    return eff * ((double)tiles / slots) / ((double)q + tail);
}

// Split K when one K loop per output tile would leave most CUs idle (the M = 3072 level: 60 tiles of 256x256).
// Returns the number of K slices (1 = do not split) for the 256x256 tile.  The slice count is a function of K ALONE
// and only the on/off decision looks at M, so that a batch and its halves (CFG sharding, tests) run the same K
// partition and stay bit-identical as long as both are in the split regime.
static inline int splitk_slices(const PncGemmParams& p) {
    if (p.geglu || p.out16t || (p.N % 256) || (p.N % 8)) return 1;
    if ((p.out32 && (p.ldc32 % 4)) || (p.out16 && (p.ldc16 % 8))) return 1;
    const long tiles = (long)((p.M + 255) / 256) * (p.N / 256);
    const int ktiles = (p.K + BK - 1) / BK;
    if (tiles > 96 || ktiles < 48) return 1;
    // (round 5: 4 slices from 56 K tiles on — the 4x48 level's temporal convs (K = 3840: 60 tiles of 256x256 x 2 slices = 120 workgroups on
    // 256 CUs, 257 TFLOP/s) and FF2 (K = 5120) fill the chip with 4; was 160)
    return ktiles >= 320 ? 8 : (ktiles >= 56 ? 4 : 2);
}

// Tile geometries.  Every channel width of the network is a multiple of 320, so the preferred tile is 256x320 (8 waves
// as 4x2, wave tile 64x160, 2 stages = 144 KB): no column waste at N = 320, A is read once per 320 output columns, 9 DMA
// instructions per 40 MFMAs.  256x256 serves GEGLU (value / gate blocks pair inside a wave) and N % 256 == 0; 256x128
// (3-stage ring) and 128x128 the small grids; 128x32 the narrow-N convs (hint stem, output head).
enum { T_128x128 = 1, T_256x128 = 2, T_256x320 = 3, T_256x256 = 4, T_128x32 = 5 };

struct TileChoice { int tile; int ksplit; };


static inline TileChoice choose_tile(const PncGemmParams& p) {
    if (p.N <= 32 && !p.geglu) return {T_128x32, 1};
    const int ks = splitk_slices(p);
    if (ks > 1 && p.ws && p.ws_floats >= (int64_t)ks * p.M * p.N && !pnc_get_option(PNC_OPT_GEMM_TILE))
        return {T_256x256, ks};
    const long mt256 = (p.M + 255) / 256, mt128 = (p.M + 127) / 128;
    const bool w320_ok = !p.geglu && (p.N % 320 == 0) && (!p.out16t || p.n_split % 320 == 0);
    const bool w256_ok = (p.N % 256 == 0) && (!p.out16t || p.n_split % 256 == 0);
    // a precise operand doubles the K loop: the long-K efficiencies apply from half the K
    const int keff = p.A_lo ? 2 * p.K : p.K;
    // measured main-loop efficiencies (relative): wide wave tiles win whenever they still fill ~3/4 of the CUs
    const double s320 = w320_ok ? tile_score(mt256 * (p.N / 320), 256, keff >= 2048 ? 1.08 : (keff >= 1024 ? 1.0 : 0.92)) : 0.0;
    const double s256 = w256_ok ? tile_score(mt256 * (p.N / 256), 256, 0.97) : 0.0;
    const double s2x1 = tile_score(mt256 * ((p.N + 127) / 128), 256, 0.80);
    const double s1x1 = tile_score(mt128 * ((p.N + 127) / 128), 512, 0.70);
    int pick = T_128x128;
    double best = s1x1;
    if (s2x1 > best) { best = s2x1; pick = T_256x128; }
    if (s256 > best) { best = s256; pick = T_256x256; }
    if (s320 > best) { best = s320; pick = T_256x320; }
    const int force = pnc_get_option(PNC_OPT_GEMM_TILE);
    if (force == T_128x128 || force == T_256x128) pick = force;
    if ((force == T_256x320 && w320_ok) || (force == T_256x256 && w256_ok)) pick = force;
    // (Round 3 measured 128x320 tiles with two 4-wave workgroups per CU — one streaming its epilogue while the other multiplies —
    // on every stream-bound shape of the path: 0-11 % slower than 256x320, profiles/round3/kbench_r3c_two_wg_128x320.log.  Not kept.)
    return {pick, 1};
}

// one workgroup owns whole output rows: the geometries that carry an E_LN variant (level-0 width 320 on 256x320; <= 128 on 128x128)
static inline bool ln_whole_rows(const PncGemmParams& p, TileChoice tc) {
    return tc.ksplit == 1 && ((tc.tile == T_256x320 && p.N <= 320) || (tc.tile == T_128x128 && p.N <= 128));
}

// launch the variant EPI of AMODE on the chosen tile
template <int AMODE, unsigned EPI>
int launch_tile(const PncGemmParams& p, hipStream_t st, TileChoice tc) {
    constexpr bool GEGLU = (EPI & E_GEGLU) != 0;
    switch (tc.tile) {
        case T_128x32:
            if constexpr (!GEGLU && !(EPI & E_VT)) return launch<AMODE, 128, 32, 4, 1, 2, true, EPI>(p, st);
            return PNC_EINVAL;
        case T_256x320:
            if constexpr (!GEGLU) return launch<AMODE, 256, 320, 4, 2, 2, false, EPI>(p, st);
            return PNC_EINVAL;
        case T_256x256:
            if (tc.ksplit > 1) return launch<AMODE, 256, 256, 4, 2, 2, true, E_O32>(p, st, tc.ksplit);   // raw partials
            return launch<AMODE, 256, 256, 4, 2, 2, true, EPI>(p, st);
        case T_256x128: return launch<AMODE, 256, 128, 4, 2, 3, true, EPI>(p, st);
        default: return launch<AMODE, 128, 128, 2, 2, 2, true, EPI>(p, st);
    }
}

}  // namespace pnc_gemm
