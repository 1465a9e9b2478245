// norm.hip — GroupNorm(32) (spatial and per-pixel temporal) + SiLU, LayerNorm.   HBM-bound.
// fp32 residual stream in (channels-last tokens), fp16 MFMA operand out, statistics in fp32
// (chunk-local sums combined with Chan's parallel-variance formula).
#include "common.h"

namespace {

constexpr int GROUPS = 32;

// Spatial GroupNorm, two launches over (pixel chunk, frame) workgroups of 4 waves.  A wave walks whole pixel rows
// (C contiguous floats): lane l owns the float4 channel vectors l, l+64, ... (J per lane), so every load is a
// fully coalesced 1 KiB wave access for any channel count, and the per-channel accumulators stay in registers.
//   stats: per-channel sums -> LDS [wave][C] -> one thread per group adds them in a FIXED order (deterministic:
//          a 1e-7 run-to-run wobble here would decorrelate the fp16 rounding of everything downstream)
//          -> partial {n, mean, M2} per (frame, chunk, group)
//   apply: Chan-combine the chunk partials, tabulate y = x*A[c] + B[c] per channel in LDS, stream.
template <int J>
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ldx, int Npix, int C,
                                                       int ppc, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [4][C]: per-wave sums, then per-wave squares
    const int f = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CV = C >> 2, cpg = C / GROUPS;
    const int p0 = chunk * ppc;
    const int p1 = min(Npix, p0 + ppc);
    f32x4 sum[J], sq[J];
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < J; ++j) { sum[j] = z; sq[j] = z; }
#pragma unroll 2
    for (int pix = p0 + wave; pix < p1; pix += 4) {
        const float* row = x + ((int64_t)f * Npix + pix) * ldx;
        f32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64;
            v[j] = (cv < CV) ? *reinterpret_cast<const f32x4*>(row + cv * 4) : z;
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            sum[j] += v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) sq[j][e] = fmaf(v[j][e], v[j][e], sq[j][e]);
        }
    }
    // two rounds through one [4][C] LDS array (sums, then squares): 16*C bytes, 40 KB at C = 2560
    float* mine = sm + wave * C;
    float ts = 0.0f, tq = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cv = lane + j * 64;
        if (cv < CV) *reinterpret_cast<f32x4*>(mine + cv * 4) = sum[j];
    }
    __syncthreads();
    if (tid < GROUPS)
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < cpg; ++c) ts += sm[w * C + tid * cpg + c];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cv = lane + j * 64;
        if (cv < CV) *reinterpret_cast<f32x4*>(mine + cv * 4) = sq[j];
    }
    __syncthreads();
    if (tid < GROUPS) {
        for (int w = 0; w < 4; ++w)
            for (int c = 0; c < cpg; ++c) tq += sm[w * C + tid * cpg + c];
        const float n = (float)(p1 - p0) * (float)cpg;
        const float mean = n > 0 ? ts / n : 0.0f;
        const float m2 = n > 0 ? fmaxf(tq - ts * mean, 0.0f) : 0.0f;
        float* o = partial + ((int64_t)(f * nchunk + chunk) * GROUPS + tid) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

template <int J>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx, int Npix, int C,
                                                       int ppc, const float* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu,
                                                       half_t* __restrict__ y, int ldy, void* __restrict__ ylo, int lo_fmt, int nrec) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // A[C], B[C], then mean[32], rstd[32]
    const int tabw = 2 * C > 24 * GROUPS ? 2 * C : 24 * GROUPS;
    float* sA = sm;
    float* sB = sm + C;
    float* s_mean = sm + tabw;
    float* s_rstd = s_mean + GROUPS;
    const int f = blockIdx.y, chunk = blockIdx.x, nchunk = nrec;       // records per frame (= the grid's chunks unless the caller says otherwise)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        // Chan-combine the chunk partials of this frame: 8 thread slices per group walk every 8th chunk (loads of a
        // slice are independent and branch-free), then one thread per group merges the 8 slices in a fixed order.
        float* s_pn = sm;                // [8][32] n, mean, M2 — scratch, overwritten by A/B afterwards
        float* s_pm = sm + 8 * GROUPS;
        float* s_p2 = sm + 16 * GROUPS;
        const int g = tid & 31, part = tid >> 5;
        float n = 0.0f, mean = 0.0f, m2 = 0.0f;
        for (int c = part; c < nchunk; c += 8) {
            const float* q = partial + ((int64_t)(f * nchunk + c) * GROUPS + g) * 3;
            const float nb = q[0], mb = q[1], m2b = q[2];
            const float nt = n + nb, d = mb - mean;
            const float w = nb / fmaxf(nt, 1.0f);
            mean = fmaf(d, w, mean);
            m2 += m2b + d * d * (n * w);
            n = nt;
        }
        s_pn[part * GROUPS + g] = n; s_pm[part * GROUPS + g] = mean; s_p2[part * GROUPS + g] = m2;
        __syncthreads();
        float fm = 0.0f, fr = 0.0f;
        if (tid < GROUPS) {
            float tn = 0.0f, tm = 0.0f, t2 = 0.0f;
            for (int k = 0; k < 8; ++k) {
                const float nb = s_pn[k * GROUPS + tid], mb = s_pm[k * GROUPS + tid], m2b = s_p2[k * GROUPS + tid];
                const float nt = tn + nb, d = mb - tm;
                const float w = nb / fmaxf(nt, 1.0f);
                tm = fmaf(d, w, tm);
                t2 += m2b + d * d * (tn * w);
                tn = nt;
            }
            fm = tm; fr = rsqrtf(t2 / tn + eps);
        }
        __syncthreads();
        if (tid < GROUPS) { s_mean[tid] = fm; s_rstd[tid] = fr; }
    }
    __syncthreads();
    const int CV = C >> 2, cpg = C / GROUPS;
    for (int c = tid; c < C; c += 256) {
        const int g = c / cpg;
        const float a = gamma[c] * s_rstd[g];
        sA[c] = a;
        sB[c] = fmaf(-s_mean[g], a, beta[c]);
    }
    __syncthreads();
    const int p0 = chunk * ppc;
    const int p1 = min(Npix, p0 + ppc);
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    // R pixel rows per wave in flight: all their loads are issued before the first use (like layernorm_kernel's LN_R)
    constexpr int R = J <= 3 ? 4 : (J <= 6 ? 2 : 1);
    for (int pix0 = p0 + wave; pix0 < p1; pix0 += 4 * R) {
        f32x4 v[R][J];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int pix = pix0 + 4 * rr;
            const float* row = x + ((int64_t)f * Npix + pix) * ldx;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int cv = lane + j * 64;
                v[rr][j] = (cv < CV && pix < p1) ? *reinterpret_cast<const f32x4*>(row + cv * 4) : z;
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int pix = pix0 + 4 * rr;
            if (pix >= p1) break;
            half_t* yrow = y + ((int64_t)f * Npix + pix) * ldy;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int cv = lane + j * 64;
                if (cv < CV) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sA + cv * 4);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sB + cv * 4);
                    half4v h;
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = fmaf(v[rr][j][e], a[e], b[e]);
                        if (silu) o[e] = silu_f(o[e]);
                        h[e] = (half_t)o[e];
                    }
                    *reinterpret_cast<half4v*>(yrow + cv * 4) = h;
                    if (ylo) store_lo4(ylo, lo_fmt, ((int64_t)f * Npix + pix) * ldy + cv * 4, o, h);
                }
            }
        }
    }
}

// ---- temporal GroupNorm + SiLU: one (b, pixel) = T rows of C --------------------------------------------------
// Block handles PB pixels; a work item = (pixel_local, 4 channels): the T float4 of an item are loaded ONCE (16-byte
// loads), kept in registers across the statistics phase and written back as 8-byte fp16 vectors (the first version
// re-read x for the second phase and moved 8 / 4 bytes per access: 3.7 TB/s).  Statistics are accumulated per channel
// PAIR in LDS (C/32 channels per group is even but not always a multiple of 4: 10 at C = 320).
// MODE 0: statistics + normalisation over the T frames of a pixel (all of them are here).  MODE 1 / 2 (round 4): the frames of a
// pixel are spread over the ranks of a frame group — 1 writes this rank's {sum, sum of squares} per (pixel, group), 2 normalises
// the local frames with the sums added over the ranks (n = values per group over ALL T_total frames); t_pad: output frame t of
// sample b goes to slot t + 1 of a (T + 2)-frame layout, the halo frames of the temporal conv around it.
template <int T, int MODE = 0>
__global__ __launch_bounds__(256) void gn_temporal_kernel(const float* __restrict__ x, int B, int Npix, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          half_t* __restrict__ y, void* __restrict__ ylo, int lo_fmt, int PB,
                                                          float* __restrict__ stats = nullptr, int T_total = T, int t_pad = 0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [PB][C/2][2] sums, then [PB][32][2] stats
    const int CP = C >> 1, C4 = C >> 2, cpg2 = (C / GROUPS) >> 1;
    float* s_part = sm;                           // PB*CP*2
    float* s_stat = sm + (size_t)PB * CP * 2;     // PB*32*2
    const int tid = threadIdx.x;
    const int64_t bp0 = (int64_t)blockIdx.x * PB;   // first (b*Npix + pixel) of the block
    const int64_t total = (int64_t)B * Npix;
    const int nwork = PB * C4;                      // <= 512 (host picks PB)
    f32x4 v[2][T];
    int64_t off[2], offo[2];
    bool live[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int wi = tid + it * 256;
        const int pl = wi / C4, c4 = wi - pl * C4;
        const int64_t bp = bp0 + pl;
        live[it] = (wi < nwork) && (bp < total);
        float s0 = 0.0f, q0 = 0.0f, s1 = 0.0f, q1 = 0.0f;
        off[it] = 0; offo[it] = 0;
        if (live[it]) {
            const int64_t b = bp / Npix, pix = bp - b * Npix;
            off[it] = ((b * T) * Npix + pix) * C + c4 * 4;
            offo[it] = ((b * (T + 2 * t_pad) + t_pad) * Npix + pix) * C + c4 * 4;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                v[it][t] = *reinterpret_cast<const f32x4*>(x + off[it] + (int64_t)t * Npix * C);
                s0 += v[it][t][0] + v[it][t][1]; q0 = fmaf(v[it][t][0], v[it][t][0], q0); q0 = fmaf(v[it][t][1], v[it][t][1], q0);
                s1 += v[it][t][2] + v[it][t][3]; q1 = fmaf(v[it][t][2], v[it][t][2], q1); q1 = fmaf(v[it][t][3], v[it][t][3], q1);
            }
        }
        if (wi < nwork) {
            float* d = s_part + ((size_t)pl * CP + c4 * 2) * 2;
            d[0] = s0; d[1] = q0; d[2] = s1; d[3] = q1;
        }
    }
    __syncthreads();
    for (int gi = tid; gi < PB * GROUPS; gi += 256) {
        const int pl = gi / GROUPS, g = gi - pl * GROUPS;
        float s = 0.0f, q = 0.0f;
        for (int j = 0; j < cpg2; ++j) {
            const int wi = pl * CP + g * cpg2 + j;
            s += s_part[wi * 2]; q += s_part[wi * 2 + 1];
        }
        if constexpr (MODE == 1) {                 // this rank's partial sums: the frame group adds them up
            const int64_t bp = bp0 + pl;
            if (bp < total) { stats[(bp * GROUPS + g) * 2] = s; stats[(bp * GROUPS + g) * 2 + 1] = q; }
            continue;
        }
        if constexpr (MODE == 2) {                 // the sums over all ranks' frames
            const int64_t bp = bp0 + pl;
            if (bp < total) { s = stats[(bp * GROUPS + g) * 2]; q = stats[(bp * GROUPS + g) * 2 + 1]; }
        }
        const float n = (float)(cpg2 * 2 * (MODE == 2 ? T_total : T));
        const float mean = s / n;
        const float var = fmaxf(q / n - mean * mean, 0.0f);
        s_stat[gi * 2] = mean; s_stat[gi * 2 + 1] = rsqrtf(var + eps);
    }
    if constexpr (MODE == 1) return;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        if (!live[it]) continue;
        const int wi = tid + it * 256;
        const int pl = wi / C4, c4 = wi - pl * C4;
        const int ga = (c4 * 2) / cpg2, gb = (c4 * 2 + 1) / cpg2;
        const float ma = s_stat[(pl * GROUPS + ga) * 2], ra = s_stat[(pl * GROUPS + ga) * 2 + 1];
        const float mb = s_stat[(pl * GROUPS + gb) * 2], rb = s_stat[(pl * GROUPS + gb) * 2 + 1];
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
        const float g0 = gm[0] * ra, g1 = gm[1] * ra, g2 = gm[2] * rb, g3 = gm[3] * rb;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float o[4] = {silu_f(fmaf(v[it][t][0] - ma, g0, bt[0])), silu_f(fmaf(v[it][t][1] - ma, g1, bt[1])),
                                silu_f(fmaf(v[it][t][2] - mb, g2, bt[2])), silu_f(fmaf(v[it][t][3] - mb, g3, bt[3]))};
            const half4v h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            *reinterpret_cast<half4v*>(y + offo[it] + (int64_t)t * Npix * C) = h;
            if (ylo) store_lo4(ylo, lo_fmt, offo[it] + (int64_t)t * Npix * C, o, h);
        }
    }
}

// ---- LayerNorm: one wave per row, LN_R rows per wave in flight (all loads issued before the first reduction),
// float4 vectors, two-pass variance in registers ------------------------------------------------------------
constexpr int LN_R = 4;
template <int J>     // J = float4 vectors per lane = ceil(C / 256)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, int M, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        half_t* __restrict__ y, int ldy, half_t* __restrict__ ylo) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_R;
    if (row0 >= M) return;
    const int CV = C >> 2;
    f32x4 v[LN_R][J];
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        const int64_t row = (row0 + r < M) ? row0 + r : (int64_t)M - 1;
        const float* xr = x + row * ldx;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64;
            v[r][j] = (cv < CV) ? *reinterpret_cast<const f32x4*>(xr + cv * 4) : z;
        }
    }
    f32x4 g[J], b[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cv = lane + j * 64;
        g[j] = (cv < CV) ? *reinterpret_cast<const f32x4*>(gamma + cv * 4) : z;
        b[j] = (cv < CV) ? *reinterpret_cast<const f32x4*>(beta + cv * 4) : z;
    }
    float mean[LN_R], rstd[LN_R];
    const float invc = 1.0f / (float)C;
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) s += (v[r][j][0] + v[r][j][1]) + (v[r][j][2] + v[r][j][3]);
        mean[r] = s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_R; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        mean[r] *= invc;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64;
            if (cv < CV) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[r][j][e] - mean[r]; q = fmaf(d, d, q); }
            }
        }
        rstd[r] = q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < LN_R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        if (row0 + r >= M) break;
        const float rs = rsqrtf(rstd[r] * invc + eps);
        half_t* yr = y + (row0 + r) * ldy;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64;
            if (cv < CV) {
                half4v h;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = fmaf((v[r][j][e] - mean[r]) * rs, g[j][e], b[j][e]); h[e] = (half_t)o[e]; }
                *reinterpret_cast<half4v*>(yr + cv * 4) = h;
                if (ylo) *reinterpret_cast<half4v*>(ylo + (row0 + r) * ldy + cv * 4) = lo_plane4(o, h);
            }
        }
    }
}

// Chan-combine the chunk records of `parts` record sets (the all-gathered partials of a view group's bands) per (frame, group):
// 8 thread slices walk every 8th record in the fixed order (set, chunk), one thread per group merges the slices — the
// arithmetic of gn_apply_kernel's own combination.  Slot 0 of the frame receives the result, the other slots an empty record.
__global__ __launch_bounds__(256) void gn_combine_kernel(const float* __restrict__ in, int parts, int F, int nchunk,
                                                         float* __restrict__ out) {
    __shared__ float s_pn[8 * GROUPS], s_pm[8 * GROUPS], s_p2[8 * GROUPS];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int g = tid & 31, part = tid >> 5;
    const int nrec = parts * nchunk;
    float n = 0.0f, mean = 0.0f, m2 = 0.0f;
    for (int r = part; r < nrec; r += 8) {
        const int s = r / nchunk, c = r - s * nchunk;
        const float* q = in + ((((int64_t)s * F + f) * nchunk + c) * GROUPS + g) * 3;
        const float nb = q[0], mb = q[1], m2b = q[2];
        const float nt = n + nb, d = mb - mean;
        const float w = nb / fmaxf(nt, 1.0f);
        mean = fmaf(d, w, mean);
        m2 += m2b + d * d * (n * w);
        n = nt;
    }
    s_pn[part * GROUPS + g] = n; s_pm[part * GROUPS + g] = mean; s_p2[part * GROUPS + g] = m2;
    __syncthreads();
    if (tid < GROUPS) {
        float tn = 0.0f, tm = 0.0f, t2 = 0.0f;
        for (int k = 0; k < 8; ++k) {
            const float nb = s_pn[k * GROUPS + tid], mb = s_pm[k * GROUPS + tid], m2b = s_p2[k * GROUPS + tid];
            const float nt = tn + nb, d = mb - tm;
            const float w = nb / fmaxf(nt, 1.0f);
            tm = fmaf(d, w, tm);
            t2 += m2b + d * d * (tn * w);
            tn = nt;
        }
        float* o = out + ((int64_t)f * nchunk * GROUPS + tid) * 3;
        o[0] = tn; o[1] = tm; o[2] = t2;
    }
    for (int i = GROUPS * 3 + tid; i < nchunk * GROUPS * 3; i += 256) out[(int64_t)f * nchunk * GROUPS * 3 + i] = 0.0f;
}

// th.cat([h, skip + control]) of the UNet's output path WITH the statistics of the GroupNorm that reads it (round 5): the concat is
// followed by the first GroupNorm(32) of a ResBlock3D on exactly the tensor it writes (openaimodel.py:1311-1314 -> 499-503), and the
// statistics launch read those C1 + C2 channels of fp32 back from HBM (level 0: 0.5-0.75 GB per site).  One workgroup = `ppc` pixels
// of one frame x ONE SLICE of the channels (32 / S whole groups: blockIdx.z): waves stride the chunk's pixels, a lane owns J <= 3
// float4 channel vectors of the slice, values are produced, stored and summed in one pass; the reduction is gn_stats_kernel's
// (per-wave channel sums in LDS, one thread per group adds them in a fixed order: bit-reproducible records).  (A first form — whole
// rows per workgroup, 64-pixel records — ran 1 wave per SIMD at C = 2560 and 48 workgroups at level 3: 3x slower than the two
// launches it replaced, profiles/round5/concat_stats_bench_r5h.log.)
template <int J>
__global__ __launch_bounds__(256) void concat_add_stats_kernel(const float* __restrict__ a, int C1, const float* __restrict__ s,
                                                               const float* __restrict__ c, int C2, int Npix, int ppc,
                                                               float* __restrict__ out32, half_t* __restrict__ out16,
                                                               void* __restrict__ out16_lo, int lo_fmt, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [4][Cs]: per-wave sums, then per-wave squares
    const int f = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x, S = gridDim.z, z = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = C1 + C2, cpg = C / GROUPS;
    const int Cs = C / S, c0 = z * Cs, CVs = Cs >> 2, gps = GROUPS / S;      // this slice: channels [c0, c0 + Cs), groups [z gps, (z + 1) gps)
    const int p0 = chunk * ppc;
    const int p1 = min(Npix, p0 + ppc);
    f32x4 sum[J], sq[J];
    const f32x4 zz = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < J; ++j) { sum[j] = zz; sq[j] = zz; }
#pragma unroll 2
    for (int pix = p0 + wave; pix < p1; pix += 4) {
        const int64_t m = (int64_t)f * Npix + pix;
        f32x4 v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64, ch = c0 + cv * 4;
            v[j] = zz;
            if (cv < CVs) {
                if (ch < C1) {
                    v[j] = *reinterpret_cast<const f32x4*>(a + m * C1 + ch);
                } else {
                    v[j] = *reinterpret_cast<const f32x4*>(s + m * C2 + (ch - C1));
                    if (c) v[j] += *reinterpret_cast<const f32x4*>(c + m * C2 + (ch - C1));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int cv = lane + j * 64, ch = c0 + cv * 4;
            if (cv < CVs) {
                if (out32) *reinterpret_cast<f32x4*>(out32 + m * C + ch) = v[j];
                if (out16) {
                    half4v h = {(half_t)v[j][0], (half_t)v[j][1], (half_t)v[j][2], (half_t)v[j][3]};
                    *reinterpret_cast<half4v*>(out16 + m * C + ch) = h;
                    if (out16_lo) {
                        const float o[4] = {v[j][0], v[j][1], v[j][2], v[j][3]};
                        store_lo4(out16_lo, lo_fmt, m * C + ch, o, h);
                    }
                }
            }
            sum[j] += v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) sq[j][e] = fmaf(v[j][e], v[j][e], sq[j][e]);
        }
    }
    float* mine = sm + wave * Cs;
    float ts = 0.0f, tq = 0.0f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cv = lane + j * 64;
        if (cv < CVs) *reinterpret_cast<f32x4*>(mine + cv * 4) = sum[j];
    }
    __syncthreads();
    if (tid < gps)
        for (int w = 0; w < 4; ++w)
            for (int cc = 0; cc < cpg; ++cc) ts += sm[w * Cs + tid * cpg + cc];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int cv = lane + j * 64;
        if (cv < CVs) *reinterpret_cast<f32x4*>(mine + cv * 4) = sq[j];
    }
    __syncthreads();
    if (tid < gps) {
        for (int w = 0; w < 4; ++w)
            for (int cc = 0; cc < cpg; ++cc) tq += sm[w * Cs + tid * cpg + cc];
        const float n = (float)(p1 - p0) * (float)cpg;
        const float mean = n > 0 ? ts / n : 0.0f;
        const float m2 = n > 0 ? fmaxf(tq - ts * mean, 0.0f) : 0.0f;
        float* o = partial + ((int64_t)(f * nchunk + chunk) * GROUPS + z * gps + tid) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

}  // namespace

constexpr int GN_MAXC = 4 * 64 * 12;       // J <= 12 float4 vectors per lane

#define PNC_GN_DISPATCH(KERNEL, ...)                                                                           \
    do {                                                                                                       \
        const int J = (C / 4 + 63) / 64;                                                                       \
        if (J <= 1) hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__);                                                \
        else if (J == 2) hipLaunchKernelGGL(KERNEL<2>, __VA_ARGS__);                                           \
        else if (J == 3) hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__);                                           \
        else if (J <= 5) hipLaunchKernelGGL(KERNEL<5>, __VA_ARGS__);                                           \
        else if (J <= 8) hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__);                                           \
        else hipLaunchKernelGGL(KERNEL<12>, __VA_ARGS__);                                                      \
    } while (0)

extern "C" int pnc_groupnorm_stats(const float* x, int ldx, int F, int Npix, int C,
                                   int pix_per_chunk, float* partial, void* stream) {
    if (!x || !partial || F < 1 || Npix < 1 || pix_per_chunk < 1) return PNC_EINVAL;
    if (C % 64 || C > GN_MAXC || ldx % 4) return PNC_EINVAL;
    if ((uintptr_t)x & 15) return PNC_EALIGN;
    const int nchunk = (Npix + pix_per_chunk - 1) / pix_per_chunk;
    const size_t lds = (size_t)4 * C * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PNC_GN_DISPATCH(gn_stats_kernel, dim3(nchunk, F), dim3(256), lds, st, x, ldx, Npix, C, pix_per_chunk, partial);
    return pnc_launch_status();
}

extern "C" int pnc_concat_add_stats(const float* a, int C1, const float* s, const float* c, int C2, int F, int Npix, int pix_per_chunk,
                                    float* out32, void* out16, void* out16_lo, int lo_fmt, float* partial, void* stream) {
    if (!a || !s || !partial || F < 1 || Npix < 1 || pix_per_chunk < 1 || C1 % 4 || C2 % 4 || C1 < 4 || C2 < 4) return PNC_EINVAL;
    if (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3) return PNC_EINVAL;
    if ((!out32 && !out16) || (out16_lo && !out16)) return PNC_EINVAL;
    const int C = C1 + C2;
    if (C % 64 || C > GN_MAXC) return PNC_EINVAL;         // (channel sums are kept per channel: a float4 vector may straddle two groups)
    if (((uintptr_t)a | (uintptr_t)s | (uintptr_t)c | (uintptr_t)out32) & 15) return PNC_EALIGN;
    const int nchunk = (Npix + pix_per_chunk - 1) / pix_per_chunk;
    // channel slices of whole groups, at most 3 float4 vectors per lane: S = 1, 2, 4 or 8
    int S = 1;
    while (S < 8 && (C / S / 4 + 63) / 64 > 3 && (C / (2 * S)) % 4 == 0) S *= 2;        // slices of whole groups AND whole float4 vectors
    if ((C / S / 4 + 63) / 64 > 3) return PNC_EINVAL;
    const int J = (C / S / 4 + 63) / 64;
    const size_t lds = (size_t)4 * (C / S) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* o16 = reinterpret_cast<half_t*>(out16);
    const dim3 grid(nchunk, F, S);
    if (J <= 1) hipLaunchKernelGGL(concat_add_stats_kernel<1>, grid, dim3(256), lds, st, a, C1, s, c, C2, Npix, pix_per_chunk, out32, o16, out16_lo, lo_fmt, partial);
    else if (J == 2) hipLaunchKernelGGL(concat_add_stats_kernel<2>, grid, dim3(256), lds, st, a, C1, s, c, C2, Npix, pix_per_chunk, out32, o16, out16_lo, lo_fmt, partial);
    else hipLaunchKernelGGL(concat_add_stats_kernel<3>, grid, dim3(256), lds, st, a, C1, s, c, C2, Npix, pix_per_chunk, out32, o16, out16_lo, lo_fmt, partial);
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_apply(const float* x, int ldx, int F, int Npix, int C,
                                   int pix_per_chunk, const float* partial,
                                   const float* gamma, const float* beta, float eps, int silu,
                                   void* y16, int ldy, void* y16_lo, int lo_fmt, int n_records, void* stream) {
    if (!x || !partial || !gamma || !beta || !y16 || F < 1 || Npix < 1 || pix_per_chunk < 1 || n_records < 0) return PNC_EINVAL;
    if (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3) return PNC_EINVAL;
    if (C % 64 || C > GN_MAXC || ldx % 4 || ldy % 4) return PNC_EINVAL;
    if (((uintptr_t)x & 15) || (((uintptr_t)y16 | (uintptr_t)y16_lo) & 7)) return PNC_EALIGN;
    const int nchunk = (Npix + pix_per_chunk - 1) / pix_per_chunk;
    const size_t lds = ((size_t)(2 * C > 24 * GROUPS ? 2 * C : 24 * GROUPS) + 2 * GROUPS) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* y = reinterpret_cast<half_t*>(y16);
    PNC_GN_DISPATCH(gn_apply_kernel, dim3(nchunk, F), dim3(256), lds, st, x, ldx, Npix, C, pix_per_chunk, partial,
                    gamma, beta, eps, silu, y, ldy, y16_lo, lo_fmt, n_records > 0 ? n_records : nchunk);
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_combine(const float* in, int parts, int F, int nchunk, float* out, void* stream) {
    if (!in || !out || parts < 1 || F < 1 || nchunk < 1) return PNC_EINVAL;
    hipLaunchKernelGGL(gn_combine_kernel, dim3(F), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, parts, F, nchunk, out);
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_temporal_silu(const float* x, int B, int T, int Npix, int C,
                                           const float* gamma, const float* beta, float eps,
                                           void* y16, void* y16_lo, int lo_fmt, void* stream) {
    if (!x || !gamma || !beta || !y16 || B < 1 || Npix < 1) return PNC_EINVAL;
    if (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3) return PNC_EINVAL;
    if (C % 64 || C > 2048 || T < 1 || T > 8) return PNC_EINVAL;     // <= 512 four-channel items per pixel
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PNC_EALIGN;
    if (((uintptr_t)y16 | (uintptr_t)y16_lo) & 7) return PNC_EALIGN;
    const int CP = C / 2;
    int PB = 512 / (C / 4); if (PB < 1) PB = 1; if (PB > 16) PB = 16;       // <= 512 work items of 4 channels per block
    const int64_t total = (int64_t)B * Npix;
    const unsigned blocks = (unsigned)((total + PB - 1) / PB);
    const size_t lds = ((size_t)PB * CP * 2 + (size_t)PB * GROUPS * 2) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* y = reinterpret_cast<half_t*>(y16);
#define PNC_GNT(TT) case TT: hipLaunchKernelGGL((gn_temporal_kernel<TT, 0>), dim3(blocks), dim3(256), lds, st, \
                                               x, B, Npix, C, gamma, beta, eps, y, y16_lo, lo_fmt, PB, (float*)nullptr, TT, 0); break;
    switch (T) {
        PNC_GNT(1) PNC_GNT(2) PNC_GNT(3) PNC_GNT(4) PNC_GNT(5) PNC_GNT(6) PNC_GNT(7) PNC_GNT(8)
    }
#undef PNC_GNT
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_temporal_part(const float* x, int B, int T, int Npix, int C,
                                           const float* gamma, const float* beta, float eps,
                                           float* stats, int mode, int T_total,
                                           void* y16, void* y16_lo, int lo_fmt, int t_pad, void* stream) {
    if (!x || !stats || B < 1 || Npix < 1 || (mode != 1 && mode != 2)) return PNC_EINVAL;
    if (C % 64 || C > 2048 || T < 1 || T > 8 || T_total < T || (t_pad != 0 && t_pad != 1)) return PNC_EINVAL;
    if (mode == 2 && (!gamma || !beta || !y16 || (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3))) return PNC_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PNC_EALIGN;
    if (((uintptr_t)y16 | (uintptr_t)y16_lo | (uintptr_t)stats) & 7) return PNC_EALIGN;
    const int CP = C / 2;
    int PB = 512 / (C / 4); if (PB < 1) PB = 1; if (PB > 16) PB = 16;
    const int64_t total = (int64_t)B * Npix;
    const unsigned blocks = (unsigned)((total + PB - 1) / PB);
    const size_t lds = ((size_t)PB * CP * 2 + (size_t)PB * GROUPS * 2) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* y = reinterpret_cast<half_t*>(y16);
#define PNC_GNTP(TT) case TT:                                                                                              \
        if (mode == 1) hipLaunchKernelGGL((gn_temporal_kernel<TT, 1>), dim3(blocks), dim3(256), lds, st, x, B, Npix, C, gamma, beta, \
                                          eps, y, y16_lo, lo_fmt, PB, stats, T_total, t_pad);                              \
        else hipLaunchKernelGGL((gn_temporal_kernel<TT, 2>), dim3(blocks), dim3(256), lds, st, x, B, Npix, C, gamma, beta, eps, y, \
                                y16_lo, lo_fmt, PB, stats, T_total, t_pad);                                                \
        break;
    switch (T) {
        PNC_GNTP(1) PNC_GNTP(2) PNC_GNTP(3) PNC_GNTP(4) PNC_GNTP(5) PNC_GNTP(6) PNC_GNTP(7) PNC_GNTP(8)
    }
#undef PNC_GNTP
    return pnc_launch_status();
}

extern "C" int pnc_layernorm(const float* x, int ldx, int M, int C,
                             const float* gamma, const float* beta, float eps,
                             void* y16, int ldy, void* y16_lo, void* stream) {
    if (!x || !gamma || !beta || !y16 || M < 1) return PNC_EINVAL;
    if (C % 4 || C > 4 * 64 * 12 || C < 4 || ldx % 4 || ldy % 4) return PNC_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PNC_EALIGN;
    if (((uintptr_t)y16 | (uintptr_t)y16_lo) & 7) return PNC_EALIGN;
    const unsigned blocks = (unsigned)(((int64_t)M + 4 * LN_R - 1) / (4 * LN_R));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* y = reinterpret_cast<half_t*>(y16);
    const int J = (C / 4 + 63) / 64;
#define PNC_LN(JJ) hipLaunchKernelGGL(layernorm_kernel<JJ>, dim3(blocks), dim3(256), 0, st, x, ldx, M, C, gamma, beta, eps, y, ldy, reinterpret_cast<half_t*>(y16_lo))
    if (J <= 1) PNC_LN(1); else if (J == 2) PNC_LN(2); else if (J == 3) PNC_LN(3); else if (J <= 5) PNC_LN(5);
    else if (J <= 8) PNC_LN(8); else PNC_LN(12);
#undef PNC_LN
    return pnc_launch_status();
}

PNC_DEFINE_TU_COLLECT(norm)
