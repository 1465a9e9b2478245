// gemm.hip — MFMA (v_mfma_f32_32x32x16_f16) GEMM family for gfx950.
//
//   C[M,N] = gatherA[M,K] x W[N,K]^T   fp16 operands, fp32 accumulate, fused epilogue.
//
// One kernel template covers every dense contraction of the Panacea denoising path
// (see include/panacea_hip.h §1 for the reference call-sites):
//   PNC_A_PLAIN     Linear / 1x1 conv on channels-last tokens
//   PNC_A_CONV3X3   implicit-GEMM 3x3 conv over an NHWC image (pad 1, stride 1|2, nearest x2 upsample)
//   PNC_A_CONV1D_T  temporal k=3 conv over the frames of one pixel
//
// Tile: BM x BN block, BK = 64, 256 threads = 4 waves, each wave owns MI x NI blocks of 32x32.
// Operands go HBM -> LDS directly (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass), 16-B
// chunks in XOR-swizzled 128-B rows; the LDS image of the DMA is lane-linear, so the swizzle is applied to
// the per-lane SOURCE address and again on the ds_read side (CDNA4 guide, rule 21).  Two LDS stages, one
// barrier per K tile: the DMA of tile t+1 is in flight while tile t feeds the MFMAs.  Out-of-range chunks
// (conv padding, K/M/N tails) are sourced from a 16-byte zero block.  The epilogue goes through LDS so that
// every global access (bias, fp32 residual stream, fp32/fp16 stores) is a 16-byte (8-byte for fp16) vector
// on 4 consecutive columns.  Workgroup ids are remapped so that each XCD walks a contiguous tile range.
#include "common.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int BK = 64;          // fp16 elements per K tile = 128 B per LDS row

// compile-time loop: the index reaches the body as a constant, so accumulator arrays are always indexed statically
// (a loop the optimizer declines to unroll would otherwise push the 256-register accumulator file to scratch)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct RowState {               // per staged A row, fixed over the K loop
    int64_t base;               // element offset of the row origin
    int y, x;                   // conv3x3: output pixel; conv1d: t in .y
    bool valid;
};

template <int AMODE>
__device__ __forceinline__ RowState make_row(const PncGemmParams& p, int m) {
    RowState s;
    s.valid = m < p.M;
    const int mm = s.valid ? m : 0;
    if (AMODE == PNC_A_PLAIN) {
        s.base = (int64_t)mm * p.lda; s.y = 0; s.x = 0;
    } else if (AMODE == PNC_A_CONV3X3) {
        const int hw = p.Hout * p.Wout;
        const int f = mm / hw, pix = mm - f * hw;
        s.y = pix / p.Wout; s.x = pix - s.y * p.Wout;
        s.base = (int64_t)f * p.Hin * p.Win * p.Cin;
    } else {
        const int f = mm / p.Npix;
        s.y = f % p.T; s.x = 0;
        s.base = (int64_t)mm * p.Cin;
    }
    return s;
}

__device__ __attribute__((aligned(16))) half_t g_zero_chunk[8];   // zero-initialised: source of padded chunks

// GEGLU gate: Phi(g) = (1 + erf(g / sqrt 2)) / 2 tabulated on [-8, 8) in steps of 1/128 as {Phi(x_i), Phi(x_{i+1}) - Phi(x_i)}
// (16 KB, copied into LDS by the GEGLU GEMMs).  Linear interpolation error <= h^2/8 max|Phi''| = 1.8e-6 — 250x below
// the fp16 rounding of the product it feeds — for 9 VALU + one ds_read_b64 per gate instead of ~14 VALU incl. exp + rcp:
// the GEGLU epilogue is VALU-issue bound and ~40 % of a K = 320 GEMM (profiles/round1/gemm_timeline_r1i.txt).
constexpr int PHI_N = 2048;
constexpr float PHI_SCALE = 128.0f, PHI_X0 = -8.0f;
constexpr int PHI_BYTES = PHI_N * 8;
__device__ __attribute__((aligned(16))) float g_phi_table[2 * PHI_N];

__device__ __forceinline__ float gelu_tab_f(float g, const float* tab) {
    float t = fmaf(g, PHI_SCALE, -PHI_X0 * PHI_SCALE);
    t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)PHI_N - 0.001f);
    const int i = (int)t;
    const float f = t - (float)i;
    const float2 e = *reinterpret_cast<const float2*>(tab + 2 * i);
    return g * fmaf(f, e.y, e.x);
}

// global source of the 16-byte chunk (row state s, k index kc) or the zero block
template <int AMODE>
__device__ __forceinline__ const half_t* a_chunk_ptr(const PncGemmParams& p, const half_t* __restrict__ A,
                                                     const RowState& s, int kc) {
    if (!s.valid || kc >= p.K) return g_zero_chunk;
    if (AMODE == PNC_A_PLAIN) {
        return A + s.base + kc;
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
        int iy, ix; bool ok;
        if (p.upsample) {
            const int uy = s.y + ky - 1, ux = s.x + kx - 1;
            ok = (uy >= 0) && (uy < p.Hout) && (ux >= 0) && (ux < p.Wout);
            iy = uy >> 1; ix = ux >> 1;
        } else {
            const int pad = p.conv_pad_br ? 0 : 1;
            iy = s.y * p.stride + ky - pad; ix = s.x * p.stride + kx - pad;
            ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        }
        return ok ? A + s.base + ((int64_t)iy * p.Win + ix) * p.Cin + ci : g_zero_chunk;
    } else {
        const int tap = kc / p.Cin, ci = kc - tap * p.Cin;
        const int tt = s.y + tap - 1;
        return (tt < 0 || tt >= p.T) ? g_zero_chunk : A + s.base + (int64_t)(tap - 1) * p.Npix * p.Cin + ci;
    }
}

// Row-major epilogue for GEMMs that READ global memory in the epilogue (residuals, per-frame row bias).  In place
// (res1 == out32) a load may not move above an earlier store, so a load-add-store loop pays one full HBM latency per
// 8-row pass (24 passes per 256x320 tile: most of a K = 320 tile's lifetime) and keeps far too few bytes in flight.
// Two measures: (1) all loads of a slab (32 rows x 64 columns) are issued together; (2) ROLLING prefetch of the first
// residual: as soon as pass ps of slab s has consumed its 8 residual values, the same registers receive the loads of
// pass ps of slab s+1 — issued before the stores of pass ps, so they travel together with those stores and under the
// LDS staging of slab s+1, at no extra VGPR cost.  Rows/columns of different slabs are disjoint, so the reordering
// is safe also when res1 aliases out32.
template <int MI, int NI>
__device__ __forceinline__ void epilogue_rowmajor_loads(const PncGemmParams& p, f32x16 (&acc)[MI][NI], float* ep, int lane,
                                                        int mw, int nw, bool ab_nostage, bool ab_nostore) {
    constexpr int ENI = NI < 2 ? NI : 2;
    constexpr int EPITCH = ENI * 32 + 4;
    constexpr int CPL = ENI * 4, RPP = 64 / CPL, NP = 32 / RPP;
    constexpr int NJ = (NI + ENI - 1) / ENI, NS = NJ * MI;
    const int cl = lane % CPL, rl = lane / CPL;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    const bool v32 = ((p.ldc32 & 3) == 0) && (((uintptr_t)p.out32 & 15) == 0);
    const bool v16 = ((p.ldc16 & 7) == 0) && (((uintptr_t)p.out16 & 15) == 0);
    const bool vr1 = ((p.ldr1 & 3) == 0) && (((uintptr_t)p.res1 & 15) == 0);
    const bool vr2 = ((p.ldr2 & 3) == 0) && (((uintptr_t)p.res2 & 15) == 0);
    const bool vrb = ((p.N & 3) == 0) && (((uintptr_t)p.rowbias & 15) == 0);
    const int act = p.act & 0xff;
    const float* xs = p.res1 ? p.res1 : p.res2;                               // stream X: the first residual
    const int ldx = p.res1 ? p.ldr1 : p.ldr2;
    const bool vx = p.res1 ? vr1 : vr2;
    const bool y_is_res2 = p.res1 && p.res2;
    const bool y_is_rb = !y_is_res2 && p.rowbias;                             // stream Y: res2, else the row bias
    const f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 x0[NP], x1[NP];
    auto load_x = [&](auto s_, auto ps_) {
        constexpr int s = decltype(s_)::value, ps = decltype(ps_)::value;
        constexpr int jc = (s / MI) * ENI, i = s % MI, cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        const int ncol = nw + jc * 32 + cl * 8;
        const int m = mw + i * 32 + ps * RPP + rl;
        x0[ps] = z4; x1[ps] = z4;
        if ((cl * 8) < cw * 32 && (ncol + 7) < p.N && xs && vx && m < p.M) {
            const float* rp = xs + (int64_t)m * ldx + ncol;
            x0[ps] = *reinterpret_cast<const f32x4*>(rp);
            x1[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
        }
    };
    static_for<NP>([&](auto ps_) { load_x(std::integral_constant<int, 0>{}, ps_); });
    static_for<NS>([&](auto s_) {
        constexpr int s = decltype(s_)::value, jc = (s / MI) * ENI, i = s % MI;
        constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;
        const int ncol = nw + jc * 32 + cl * 8;
        const bool lane_on = (cl * 8) < cw * 32 && ncol < p.N;
        const bool full8 = (ncol + 7) < p.N;
        // this slab's second stream and bias
        f32x4 y0[NP], y1[NP];
        float bcol[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bcol[e] = (p.bias && lane_on && (ncol + e) < p.N) ? p.bias[ncol + e] : 0.0f;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int m = mw + i * 32 + ps * RPP + rl;
            const bool ok = lane_on && full8 && m < p.M;
            y0[ps] = z4; y1[ps] = z4;
            if (ok && y_is_res2 && vr2) {
                const float* rp = p.res2 + (int64_t)m * p.ldr2 + ncol;
                y0[ps] = *reinterpret_cast<const f32x4*>(rp);
                y1[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
            }
            if (ok && y_is_rb && vrb) {
                const float* rp = p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
                y0[ps] = *reinterpret_cast<const f32x4*>(rp);
                y1[ps] = *reinterpret_cast<const f32x4*>(rp + 4);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // previous slab fully read back from LDS
        static_for<cw>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if (!ab_nostage) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)] = acc[i][jc + j][r];
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<NP>([&](auto ps_) {
            constexpr int ps = decltype(ps_)::value;
            const float* src = ep + (ps * RPP + rl) * EPITCH + cl * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
            const int m = mw + i * 32 + ps * RPP + rl;
            const bool row_on = lane_on && m < p.M;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[e] + bcol[e]; v[e + 4] = a1[e] + bcol[e + 4]; }
            if (p.rowbias && row_on) {
                if (y_is_rb && full8 && vrb) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += y0[ps][e]; v[e + 4] += y1[ps][e]; }
                } else {
                    const float* rb = p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) v[e] += rb[e];
                }
            }
            if (act == PNC_ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
            }
            if (xs && row_on) {                 // first residual (res1, or res2 when it is the only one)
                if (full8 && vx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += x0[ps][e]; v[e + 4] += x1[ps][e]; }
                } else {
                    const float* rp = xs + (int64_t)m * ldx + ncol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) v[e] += rp[e];
                }
            }
            // rolling prefetch: these registers are free now; the loads go out before this pass's stores
            if constexpr (s + 1 < NS) load_x(std::integral_constant<int, s + 1>{}, ps_);
            if (y_is_res2 && row_on) {
                if (full8 && vr2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += y0[ps][e]; v[e + 4] += y1[ps][e]; }
                } else {
                    const float* rp = p.res2 + (int64_t)m * p.ldr2 + ncol;
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) v[e] += rp[e];
                }
            }
            if (!row_on) return;
            if (ab_nostore) { if (v[0] == 123.456f) p.out32[0] = v[1] + v[5]; return; }
            if (p.out32) {
                float* op = p.out32 + (int64_t)m * p.ldc32 + ncol;
                if (full8 && v32) {
                    f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(op) = o0;
                    *reinterpret_cast<f32x4*>(op + 4) = o1;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) op[e] = v[e];
                }
            }
            if (out16) {
                half_t* op = out16 + (int64_t)m * p.ldc16 + ncol;
                if (full8 && v16) {
                    half8v o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                    *reinterpret_cast<half8v*>(op) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) op[e] = (half_t)v[e];
                }
            }
        });
    });
}

// Row-major epilogue of one wave tile (MI x NI blocks of 32x32).  Each 32-row x 64-column slab goes through a
// wave-private LDS region so that a lane ends up with 8 CONSECUTIVE columns of one row: 16-byte fp16 stores, two
// 16-byte fp32 loads/stores.  All LDS reads of a slab are issued before the first use (the loop is instruction- and
// latency-bound, not bandwidth-bound: at K = 320 it is half of a tile's lifetime).  LDS operations of one wave
// execute in order, so only lgkmcnt waits separate the phases — no workgroup barrier.
template <int MI, int NI, bool GEGLU>
__device__ __forceinline__ void epilogue_rowmajor(const PncGemmParams& p, f32x16 (&acc)[MI][NI], float* ep, int lane,
                                                  int mw, int nw, bool ab_nostage, bool ab_nostore,
                                                  const float* phi_tab) {
    constexpr int ENI = NI < 2 ? NI : 2;
    constexpr int EPITCH = ENI * 32 + 4;
    constexpr int OUTC = GEGLU ? 32 : ENI * 32;             // output columns of one staged chunk
    constexpr int CPL = OUTC / 8;                           // lanes per row (8 columns per lane)
    constexpr int RPP = 64 / CPL;                           // rows per pass
    constexpr int NP = 32 / RPP;                            // passes per 32-row slab
    const int cl = lane % CPL, rl = lane / CPL;
    const int Nout = GEGLU ? (p.N >> 1) : p.N;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    const bool v32 = ((p.ldc32 & 3) == 0) && (((uintptr_t)p.out32 & 15) == 0);
    const bool v16 = ((p.ldc16 & 7) == 0) && (((uintptr_t)p.out16 & 15) == 0);
    const bool vr1 = ((p.ldr1 & 3) == 0) && (((uintptr_t)p.res1 & 15) == 0);
    const bool vr2 = ((p.ldr2 & 3) == 0) && (((uintptr_t)p.res2 & 15) == 0);
    const int act = p.act & 0xff;
    if constexpr (!GEGLU) {
        if (p.res1 || p.res2 || p.rowbias) {
            epilogue_rowmajor_loads<MI, NI>(p, acc, ep, lane, mw, nw, ab_nostage, ab_nostore);
            return;
        }
    }
    static_for<(NI + ENI - 1) / ENI>([&](auto jc_) {
        constexpr int jc = decltype(jc_)::value * ENI;
        constexpr int cw = (NI - jc) < ENI ? (NI - jc) : ENI;           // column blocks in this chunk (1 or 2)
        const int nin0 = nw + jc * 32 + cl * 8;                         // first input column (N space of W / bias)
        const int ncol = GEGLU ? ((nw + jc * 32) >> 1) + cl * 8 : nin0; // first output column of this lane
        const bool lane_on = (cl * 8) < (GEGLU ? 32 : cw * 32) && ncol < Nout;
        const bool full8 = (ncol + 7) < Nout;
        float bcol[8], bgate[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int nin = nin0 + e;
            bcol[e] = (p.bias && lane_on && nin < p.N) ? p.bias[nin] : 0.0f;
            bgate[e] = (GEGLU && p.bias && lane_on && (nin + 32) < p.N) ? p.bias[nin + 32] : 0.0f;
        }
        static_for<MI>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // previous slab fully read
            static_for<cw>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                if (!ab_nostage) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        ep[mfma32_row(r, lane) * EPITCH + j * 32 + (lane & 31)] = acc[i][jc + j][r];
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 a0[NP], a1[NP], g0[NP], g1[NP];
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const float* src = ep + (ps * RPP + rl) * EPITCH + cl * 8;
                a0[ps] = *reinterpret_cast<const f32x4*>(src);
                a1[ps] = *reinterpret_cast<const f32x4*>(src + 4);
                if (GEGLU) {
                    g0[ps] = *reinterpret_cast<const f32x4*>(src + 32);
                    g1[ps] = *reinterpret_cast<const f32x4*>(src + 36);
                }
            }
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                const int m = mw + i * 32 + ps * RPP + rl;
                if (!lane_on || m >= p.M) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a0[ps][e] + bcol[e]; v[e + 4] = a1[ps][e] + bcol[e + 4]; }
                if (GEGLU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] *= gelu_tab_f(g0[ps][e] + bgate[e], phi_tab);
                        v[e + 4] *= gelu_tab_f(g1[ps][e] + bgate[e + 4], phi_tab);
                    }
                } else {
                    if (p.rowbias) {
                        const float* rb = p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (ncol + e < Nout) v[e] += rb[e];
                    }
                    if (act == PNC_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                    }
                    if (p.res1) {
                        const float* rp = p.res1 + (int64_t)m * p.ldr1 + ncol;
                        if (full8 && vr1) {
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (ncol + e < Nout) v[e] += rp[e];
                        }
                    }
                    if (p.res2) {
                        const float* rp = p.res2 + (int64_t)m * p.ldr2 + ncol;
                        if (full8 && vr2) {
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[e + 4] += r1[e]; }
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (ncol + e < Nout) v[e] += rp[e];
                        }
                    }
                }
                if (ab_nostore) { if (v[0] == 123.456f) p.out32[0] = v[1] + v[5]; continue; }
                if (p.out32) {
                    float* op = p.out32 + (int64_t)m * p.ldc32 + ncol;
                    if (full8 && v32) {
                        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                        *reinterpret_cast<f32x4*>(op) = o0;
                        *reinterpret_cast<f32x4*>(op + 4) = o1;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (ncol + e < Nout) op[e] = v[e];
                    }
                }
                if (out16) {
                    half_t* op = out16 + (int64_t)m * p.ldc16 + ncol;
                    if (full8 && v16) {
                        half8v o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                        *reinterpret_cast<half8v*>(op) = o;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (ncol + e < Nout) op[e] = (half_t)v[e];
                    }
                }
            }
        });
    });
}

template <int AMODE, int BM, int BN, int WGM, int WGN, int STAGES, bool PIPE>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_glds_kernel(const PncGemmParams pin, const int ksplit,
                                                                   const int nfull, const int tail_f) {
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
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    // split K (ksplit > 1): block b = (slice, tile); slice s runs K tiles [s*nt/S, (s+1)*nt/S) and writes its raw fp32
    // accumulators to ws[s][M][N]; splitk_reduce_kernel sums the slices in order and applies the epilogue
    // Tail split (tail_f = 2 or 4): the last (ntile_mn - nfull) output tiles - the partial round that would leave most
    // CUs idle - are each run by tail_f workgroups that own BM / tail_f rows of the tile: the waves of the other row
    // groups skip their MFMAs and epilogue (their A rows are DMA'd as zero chunks), all waves still stage W.  Rows are
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
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ntiles_all = (p.K + BK - 1) / BK;
    const int kt_begin = (int)((int64_t)kslice * ntiles_all / ksplit);
    const int ntiles = (int)((int64_t)(kslice + 1) * ntiles_all / ksplit) - kt_begin;
    if (ksplit > 1) {            // raw partial sums; the reduce launch owns bias / residuals / outputs
        p.out32 = p.ws + (int64_t)kslice * p.M * p.N;
        p.ldc32 = p.N;
        p.bias = nullptr; p.rowbias = nullptr; p.res1 = nullptr; p.res2 = nullptr;
        p.out16 = nullptr; p.out16t = nullptr; p.n_split = p.N; p.act &= ~0xff;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        rows[i] = make_row<AMODE>(p, m0 + r);
        rows[i].valid = rows[i].valid && (r >= rows_lo) && (r < rows_hi);
    }
    const half_t* wrow[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + i * RPI + srow;
        wrow[i] = (n < p.N) ? Wt + (int64_t)n * p.K : nullptr;
    }
    auto issue_tile = [&](int kt_local, int stage) {
        const int kt = kt_begin + kt_local;
        const int kc = kt * BK + schunk * 8;
        char* sa = smem + stage * STAGE + wave * 1024;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16(a_chunk_ptr<AMODE>(p, A, rows[i], kc), sa + i * (RPI * 128));
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            glds16((wrow[i] && kc < p.K) ? wrow[i] + kc : g_zero_chunk, sb + i * (RPI * 128));
    };

    // GEGLU: the Phi table rides into LDS (behind the operand ring) with the first K tile
    constexpr int RING_BYTES = STAGES * STAGE;
    if (p.geglu) {
#pragma unroll
        for (int c = wave; c < PHI_BYTES / 1024; c += NW)
            glds16(reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(g_phi_table) + c * 1024 + lane * 16),
                   smem + RING_BYTES + c * 1024);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fk = lane >> 5;
    // timing-experiment switches (tools/kbench.py PNC_ABLATE): results are garbage when any is set
    const bool ab_nodma = (p.act & 0x100) != 0, ab_nomfma = (p.act & 0x200) != 0, ab_noepi = (p.act & 0x400) != 0;
    const bool ab_nostage = (p.act & 0x800) != 0, ab_nostore = (p.act & 0x1000) != 0, ab_nobar = (p.act & 0x2000) != 0;
    const bool prio = (p.act & 0x4000) != 0;
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
                if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
                if (prio) __builtin_amdgcn_s_setprio(0);
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

    if (STAGES == 2) {
        // one tile in flight: the plain barrier carries the vmcnt(0) that lands the DMA
        issue_tile(0, 0);
        __syncthreads();
        // The second-dispatched half of the waves (4-7: one per SIMD, the arbitration losers) issues its share of the
        // next tile's DMA in the MIDDLE of its MFMA stream instead of together with waves 0-3 right after the barrier
        // (s_memtime timeline: 1870 vs 690 cycles per tile in the issue segment, with waves 0-3 then idling ~1400
        // cycles at the barrier): each SIMD then has one wave issuing DMA while the other runs MFMAs.
        const bool late = ((p.act & 0x40000) == 0) && NW == 8 && wave >= 4 && wave_on && ntiles >= 8;   // 2-5 % at long K
        for (int kt = 0; kt < ntiles; ++kt) {
            const bool nxt = kt + 1 < ntiles && !ab_nodma;
            if (nxt && !late) issue_tile(kt + 1, (kt + 1) & 1);
            if (!ab_nomfma && wave_on) compute(kt & 1, (nxt && late) ? kt + 1 : -1);
            if (!ab_nobar) __syncthreads();
        }
    } else {
        // ring of three stages, TWO tiles in flight.  Counted waits: after issuing tile kt+2 only its LOADS
        // DMA instructions may stay outstanding, i.e. tile kt+1 has landed; the raw s_barrier (no compiler
        // vmcnt(0)) then publishes every wave's part of it and retires all reads of the stage being recycled.
        issue_tile(0, 0);
        if (ntiles > 1) {
            issue_tile(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        int st = 0;
        for (int kt = 0; kt < ntiles; ++kt) {
            const bool ahead = (kt + 2) < ntiles && !ab_nodma;
            if (ahead) issue_tile(kt + 2, st == 0 ? 2 : st - 1);      // (kt + 2) % 3
            if (!ab_nomfma && wave_on) compute(st);
            if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            st = (st == 2) ? 0 : st + 1;
        }
    }

    // ------------------------------ epilogue ------------------------------
    if (ab_noepi) { if (acc[0][0][0] == 123.456f) p.out32[0] = 1.0f; return; }
    if (!wave_on) return;                       // row group of another workgroup (tail split)
    const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    half_t* out16t = reinterpret_cast<half_t*>(p.out16t);

    if ((out16t != nullptr) && (n0 >= p.n_split)) {
        // channel-major ("V^T") output: a lane already holds 4 consecutive rows of one column
        const int col = lane & 31;
        static_for<MI * NI>([&](auto ij_) {
            constexpr int i = decltype(ij_)::value / NI, j = decltype(ij_)::value % NI;
            {
                const int n = nw + j * 32 + col;
                if (n >= p.N) return;
                const float bn = p.bias ? p.bias[n] : 0.0f;
                const int grp = lane >> 5;
                const bool al16 = ((p.ldt & 7) == 0) && ((p.t_gstride & 7) == 0) && (((uintptr_t)p.out16t & 15) == 0);
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    // rows 8*r4 + {0..3} live in lane l, {4..7} in lane l+32 (same column): one xor-32 exchange per
                    // pair of r4 gives each lane 8 consecutive rows -> 16-byte stores along the token axis of V^T
                    half4v he, ho;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        he[q] = (half_t)(acc[i][j][(2 * rp) * 4 + q] + bn);
                        ho[q] = (half_t)(acc[i][j][(2 * rp + 1) * 4 + q] + bn);
                    }
                    const int m8a = mw + i * 32 + 8 * (2 * rp), m8b = m8a + 8;       // both 8-row groups of the pair
                    const int ga = m8a / p.t_rows, ta = m8a - ga * p.t_rows;
                    const int gb = m8b / p.t_rows, tb = m8b - gb * p.t_rows;
                    const bool fast = al16 && (m8b + 7 < p.M) && (ta + 7 < p.t_rows) && (tb + 7 < p.t_rows) &&
                                      ((ta & 7) == 0) && ((tb & 7) == 0);
                    if (fast) {
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
                        half_t* dst = out16t + (int64_t)(grp ? gb : ga) * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt +
                                      (grp ? tb : ta);
                        *reinterpret_cast<half8v*>(dst) = o8;
                    } else {
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            const half4v h = rr ? ho : he;
                            const int mb = mw + i * 32 + 8 * (2 * rp + rr) + 4 * grp;
                            const int g = mb / p.t_rows, tr = mb - g * p.t_rows;
                            half_t* dst = out16t + (int64_t)g * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + tr;
                            const bool vec = (mb + 3 < p.M) && (tr + 3 < p.t_rows) && ((p.ldt & 3) == 0) &&
                                             ((p.t_gstride & 3) == 0) && ((tr & 3) == 0);
                            if (vec) {
                                *reinterpret_cast<half4v*>(dst) = h;
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int m = mb + q;
                                    if (m < p.M) {
                                        const int g2 = m / p.t_rows, t2 = m - g2 * p.t_rows;
                                        out16t[(int64_t)g2 * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + t2] = h[q];
                                    }
                                }
                            }
                        }
                    }
                }
            }
        });
        return;
    }

    // row-major outputs
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EPITCH);
    __syncthreads();                        // every wave is done reading operand tiles from LDS
    if (p.geglu) {
        if constexpr (NI >= 2)
            epilogue_rowmajor<MI, NI, true>(p, acc, ep, lane, mw, nw, ab_nostage, ab_nostore,
                                            reinterpret_cast<const float*>(smem + RING_BYTES));
    } else {
        epilogue_rowmajor<MI, NI, false>(p, acc, ep, lane, mw, nw, ab_nostage, ab_nostore, nullptr);
    }
}

// One-time upload of the Phi table (host-computed in double).  Synchronous, hence not legal during stream capture: the
// first GEGLU GEMM of a process has to run eagerly (every warm-up does).
static int ensure_phi_table(hipStream_t st) {
    static bool ready_dev[64] = {};                  // the table is a per-device symbol
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& ready = ready_dev[dev & 63];
    if (ready) return PNC_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return PNC_EINVAL;
    static float host[2 * PHI_N];
    auto phi = [](double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)); };
    for (int i = 0; i < PHI_N; ++i) {
        const double x0 = (double)PHI_X0 + (double)i / PHI_SCALE, x1 = (double)PHI_X0 + (double)(i + 1) / PHI_SCALE;
        host[2 * i] = (float)phi(x0);
        host[2 * i + 1] = (float)(phi(x1) - phi(x0));
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phi_table), host, sizeof(host)) != hipSuccess) return (int)hipGetLastError();
    ready = true;
    return PNC_OK;
}

// Workgroups of one geometry that are resident at once on the 256 CUs (LDS-limited: 160 KB per CU)
template <int LDS_BYTES>
constexpr int resident_slots() { return 256 * ((160 * 1024) / LDS_BYTES < 1 ? 1 : (160 * 1024) / LDS_BYTES > 2 ? 2 : (160 * 1024) / LDS_BYTES); }

// Tail split decision: tiles = q * slots + r.  When the last, partial round holds r <= slots/2 (or /4) tiles, run each of
// them as 2 (4) workgroups of BM/2 (BM/4) rows so that the round fills the chip: e.g. M = 49152, N = 640 with 256x320
// tiles is 384 tiles = 1.5 rounds -> 256 full tiles + 128 tiles x 2 halves.
template <int BM, int WGM, int LDS_BYTES>
static inline void tail_split(int tiles, int& nfull, int& tail_f) {
    const bool off = getenv("PNC_GEMM_NOTAIL") != nullptr;             // A/B runs, tests
    constexpr int slots = resident_slots<LDS_BYTES>();
    const int r = tiles % slots;
    nfull = tiles; tail_f = 1;
    if (off || r == 0) return;
    if (WGM >= 4 && r * 4 <= slots) tail_f = 4;
    else if (WGM >= 2 && r * 2 <= slots) tail_f = 2;
    if (tail_f > 1) nfull = tiles - r;
}

// Second launch of a split-K GEMM: out = epilogue(sum_s ws[s]) with the slices summed in index order (deterministic).
// One lane owns 8 consecutive columns of one row (N % 8 == 0 is a precondition of splitting).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const PncGemmParams p, const int ksplit) {
    const int n8 = p.N >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * n8) return;
    const int m = (int)(idx / n8), ncol = (int)(idx - (int64_t)m * n8) * 8;
    const int64_t mn = (int64_t)p.M * p.N;
    const float* src = p.ws + (int64_t)m * p.N + ncol;
    f32x4 s0 = *reinterpret_cast<const f32x4*>(src), s1 = *reinterpret_cast<const f32x4*>(src + 4);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + s * mn), t1 = *reinterpret_cast<const f32x4*>(src + s * mn + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] += t0[e]; s1[e] += t1[e]; }
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = s0[e]; v[e + 4] = s1[e]; }
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[ncol + e];
    }
    if (p.rowbias) {
        const float* rb = p.rowbias + (int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + ncol;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rb[e];
    }
    if ((p.act & 0xff) == PNC_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (p.res1) {
        const float* rp = p.res1 + (int64_t)m * p.ldr1 + ncol;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rp[e];
    }
    if (p.res2) {
        const float* rp = p.res2 + (int64_t)m * p.ldr2 + ncol;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rp[e];
    }
    if (p.out32) {
        float* op = p.out32 + (int64_t)m * p.ldc32 + ncol;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = v[e];
    }
    if (p.out16) {
        half_t* op = reinterpret_cast<half_t*>(p.out16) + (int64_t)m * p.ldc16 + ncol;
#pragma unroll
        for (int e = 0; e < 8; ++e) op[e] = (half_t)v[e];
    }
}

template <int AMODE, int BM, int BN, int WGM, int WGN, int STAGES, bool PIPE>
int launch(const PncGemmParams& p, hipStream_t st, int ksplit = 1) {
    constexpr int lds = STAGES * (BM + BN) * 128;
    constexpr int threads = 64 * WGM * WGN;
    static_assert(lds + PHI_BYTES <= 160 * 1024, "LDS budget of one CU (operand ring + GEGLU table)");
    static_assert(lds >= WGM * WGN * 32 * 68 * 4, "epilogue staging must fit the operand ring");
    static bool attr_done_dev[64] = {};   // per instantiation and device; idempotent
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool& attr_done = attr_done_dev[dev & 63];
    auto kern = gemm_glds_kernel<AMODE, BM, BN, WGM, WGN, STAGES, PIPE>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds + PHI_BYTES);
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    int nfull = tiles, tail_f = 1;
    if (ksplit == 1) tail_split<BM, WGM, lds>(tiles, nfull, tail_f);
    const int blocks = ksplit > 1 ? tiles * ksplit : nfull + (tiles - nfull) * tail_f;
    PncGemmParams q = p;
    if (getenv("PNC_GEMM_NOLATE")) q.act |= 0x40000;         // A/B runs: all waves issue their DMA share after the barrier
    if (p.geglu) {
        const int rc = ensure_phi_table(st);
        if (rc != PNC_OK) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds + (p.geglu ? PHI_BYTES : 0), st, q, ksplit, nfull, tail_f);
    if (ksplit > 1) {
        const int64_t work = (int64_t)p.M * (p.N >> 3);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, p, ksplit);
    }
    return pnc_launch_status();
}

// Tile choice.  Every channel width of the network is a multiple of 320, so the preferred tile is 256x320
// (8 waves as 4x2, wave tile 64x160, 2 stages = 144 KB): no column waste at N = 320, A is read once per 320
// output columns, 9 DMA instructions per 40 MFMAs (vs 6 per 16 for 256x128).  GEGLU pairs 32-column value /
// gate blocks inside one wave and therefore keeps 256x128 (wave tile 64x64).  Small grids fall back to
// 128x128 so that every CU still gets work; 128x32 serves the narrow-N convs (hint stem, output head).
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
    return ktiles >= 320 ? 8 : (ktiles >= 160 ? 4 : 2);
}

template <int AMODE>
int dispatch(const PncGemmParams& p, hipStream_t st) {
    static const int force = getenv("PNC_GEMM_TILE") ? atoi(getenv("PNC_GEMM_TILE")) : 0;   // A/B runs
    if (p.N <= 32 && !p.geglu) return launch<AMODE, 128, 32, 4, 1, 2, true>(p, st);
    if (!force || force == 7) {
        const int ks = splitk_slices(p);
        if (ks > 1 && p.ws && p.ws_floats >= (int64_t)ks * p.M * p.N)
            return launch<AMODE, 256, 256, 4, 2, 2, true>(p, st, ks);
    }
    const long mt256 = (p.M + 255) / 256, mt128 = (p.M + 127) / 128;
    const bool w320_ok = !p.geglu && (p.N % 320 == 0) && (!p.out16t || p.n_split % 320 == 0);
    const bool w256_ok = (p.N % 256 == 0) && (!p.out16t || p.n_split % 256 == 0);
    int pick = force;
    if (!pick) {
        // measured main-loop efficiencies (relative): wide wave tiles win whenever they still fill ~3/4 of the CUs
        const double s320 = w320_ok ? tile_score(mt256 * (p.N / 320), 256, p.K >= 2048 ? 1.08 : (p.K >= 1024 ? 1.0 : 0.92)) : 0.0;
        const double s256 = w256_ok ? tile_score(mt256 * (p.N / 256), 256, 0.97) : 0.0;
        const double s2x1 = tile_score(mt256 * ((p.N + 127) / 128), 256, 0.80);
        const double s1x1 = tile_score(mt128 * ((p.N + 127) / 128), 512, 0.70);
        pick = 1;
        double best = s1x1;
        if (s2x1 > best) { best = s2x1; pick = 2; }
        if (s256 > best) { best = s256; pick = 4; }
        if (s320 > best) { best = s320; pick = 3; }
    }
    if (pick == 3 && w320_ok) return launch<AMODE, 256, 320, 4, 2, 2, false>(p, st);
    if (pick == 4 && w256_ok) return launch<AMODE, 256, 256, 4, 2, 2, true>(p, st);
    if (pick == 2) return launch<AMODE, 256, 128, 4, 2, 3, true>(p, st);
    return launch<AMODE, 128, 128, 2, 2, 2, true>(p, st);
}

}  // namespace

extern "C" int64_t pnc_gemm_workspace_floats(const PncGemmParams* pp) {
    if (!pp || pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
    if (pp->N <= 32 && !pp->geglu) return 0;
    const int ks = splitk_slices(*pp);
    return ks > 1 ? (int64_t)ks * pp->M * pp->N : 0;
}

extern "C" int pnc_gemm_f16(const PncGemmParams* pp, void* stream) {
    if (!pp) return PNC_EINVAL;
    const PncGemmParams& p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.W) return PNC_EINVAL;
    if (p.K % 8) return PNC_EINVAL;                       // 16-byte operand chunks
    if (((uintptr_t)p.A | (uintptr_t)p.W) & 15) return PNC_EALIGN;
    if (p.a_mode == PNC_A_PLAIN) {
        if (p.lda % 8 || p.lda < p.K) return PNC_EALIGN;
    } else if (p.a_mode == PNC_A_CONV3X3) {
        if (p.Cin % 8 || p.K != 9 * p.Cin || p.Hin <= 0 || p.Win <= 0) return PNC_EINVAL;
        if (p.stride != 1 && p.stride != 2) return PNC_EINVAL;
        if (p.upsample && (p.stride != 1 || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win)) return PNC_EINVAL;
        if (p.conv_pad_br && (p.upsample || p.stride != 2)) return PNC_EINVAL;
        if (p.M % (p.Hout * p.Wout)) return PNC_EINVAL;
    } else if (p.a_mode == PNC_A_CONV1D_T) {
        if (p.Cin % 8 || p.K != 3 * p.Cin || p.T <= 0 || p.Npix <= 0) return PNC_EINVAL;
        if (p.M % (p.T * p.Npix)) return PNC_EINVAL;
    } else {
        return PNC_EINVAL;
    }
    if (p.rowbias && (p.rb_rows <= 0 || p.rb_mod <= 0)) return PNC_EINVAL;
    if (p.geglu && ((p.N % 64) || p.out16t || p.res1 || p.res2 || p.rowbias)) return PNC_EINVAL;
    if (p.out16t && ((p.n_split % 128) || p.t_rows <= 0 || p.N <= 32)) return PNC_EINVAL;
    if (!p.out32 && !p.out16 && !p.out16t) return PNC_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    switch (p.a_mode) {
        case PNC_A_PLAIN: return dispatch<PNC_A_PLAIN>(p, st);
        case PNC_A_CONV3X3: return dispatch<PNC_A_CONV3X3>(p, st);
        default: return dispatch<PNC_A_CONV1D_T>(p, st);
    }
}
