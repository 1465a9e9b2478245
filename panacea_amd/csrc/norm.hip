// norm.hip — GroupNorm(32) (spatial and per-pixel temporal) + SiLU, LayerNorm.   HBM-bound.
// fp32 residual stream in (channels-last tokens), fp16 MFMA operand out, statistics in fp32
// (chunk-local sums combined with Chan's parallel-variance formula).
#include "common.h"

namespace {

constexpr int GROUPS = 32;
constexpr int MAXV = 3;        // float4 channel vectors per thread: C <= 4*256*3 = 3072

// thread t owns channel vectors cv = (t % CVT) + j*CVT  and pixel lane pl = t / CVT
struct ChanMap { int CV, CVT, PL, cv0, pl; bool active; };
__device__ __forceinline__ ChanMap chan_map(int C, int tid) {
    ChanMap m;
    m.CV = C >> 2;
    m.CVT = m.CV < 256 ? m.CV : 256;
    m.PL = 256 / m.CVT;
    m.cv0 = tid % m.CVT;
    m.pl = tid / m.CVT;
    m.active = m.pl < m.PL;
    return m;
}

__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ldx, int Npix, int C,
                                                       int ppc, float* __restrict__ partial) {
    // Deterministic: per-thread channel sums go to LDS [pixel lane][channel] and ONE thread per group adds
    // them in a fixed order (no float atomics: a 1e-7 run-to-run wobble here would decorrelate the fp16
    // rounding of everything downstream).
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [PL][C] sums, then [PL][C] squares
    const int f = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int tid = threadIdx.x;
    const ChanMap cm = chan_map(C, tid);
    const int cpg = C / GROUPS;
    const int p0 = chunk * ppc;
    const int p1 = min(Npix, p0 + ppc);
    float* s_sum = sm;
    float* s_sq = sm + cm.PL * C;
    float sum[MAXV][4], sq[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sum[j][e] = 0.0f; sq[j][e] = 0.0f; }
    if (cm.active) {
        for (int pix = p0 + cm.pl; pix < p1; pix += cm.PL) {
            const float* row = x + ((int64_t)f * Npix + pix) * ldx;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int cv = cm.cv0 + j * cm.CVT;
                if (cv < cm.CV) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(row + cv * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sum[j][e] += v[e]; sq[j][e] = fmaf(v[e], v[e], sq[j][e]); }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int cv = cm.cv0 + j * cm.CVT;
            if (cv < cm.CV) {
                f32x4 a = {sum[j][0], sum[j][1], sum[j][2], sum[j][3]};
                f32x4 b = {sq[j][0], sq[j][1], sq[j][2], sq[j][3]};
                *reinterpret_cast<f32x4*>(s_sum + cm.pl * C + cv * 4) = a;
                *reinterpret_cast<f32x4*>(s_sq + cm.pl * C + cv * 4) = b;
            }
        }
    }
    __syncthreads();
    if (tid < GROUPS) {
        float ts = 0.0f, tq = 0.0f;
        for (int pl = 0; pl < cm.PL; ++pl)
            for (int c = 0; c < cpg; ++c) {
                ts += s_sum[pl * C + tid * cpg + c];
                tq += s_sq[pl * C + tid * cpg + c];
            }
        const float n = (float)(p1 - p0) * (float)cpg;
        const float mean = n > 0 ? ts / n : 0.0f;
        const float m2 = n > 0 ? fmaxf(tq - ts * mean, 0.0f) : 0.0f;
        float* o = partial + ((int64_t)(f * nchunk + chunk) * GROUPS + tid) * 3;
        o[0] = n; o[1] = mean; o[2] = m2;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ldx, int Npix, int C,
                                                       int ppc, const float* __restrict__ partial,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int silu,
                                                       half_t* __restrict__ y, int ldy) {
    __shared__ float s_mean[GROUPS], s_rstd[GROUPS];
    const int f = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int tid = threadIdx.x;
    if (tid < GROUPS) {
        float n = 0.0f, mean = 0.0f, m2 = 0.0f;
        for (int c = 0; c < nchunk; ++c) {
            const float* q = partial + ((int64_t)(f * nchunk + c) * GROUPS + tid) * 3;
            const float nb = q[0], mb = q[1], m2b = q[2];
            if (nb > 0.0f) {
                const float nt = n + nb, d = mb - mean;
                mean += d * (nb / nt);
                m2 += m2b + d * d * (n * nb / nt);
                n = nt;
            }
        }
        s_mean[tid] = mean;
        s_rstd[tid] = rsqrtf(m2 / n + eps);
    }
    __syncthreads();
    const ChanMap cm = chan_map(C, tid);
    if (!cm.active) return;
    const int cpg = C / GROUPS;
    const int p0 = chunk * ppc;
    const int p1 = min(Npix, p0 + ppc);
    float ga[MAXV][4], be[MAXV][4], mu[MAXV][4];
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = cm.cv0 + j * cm.CVT;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (cv < cm.CV) {
                const int c = cv * 4 + e, gidx = c / cpg;
                const float g = gamma[c] * s_rstd[gidx];
                ga[j][e] = g; mu[j][e] = s_mean[gidx]; be[j][e] = beta[c];
            } else { ga[j][e] = 0.0f; mu[j][e] = 0.0f; be[j][e] = 0.0f; }
        }
    }
    for (int pix = p0 + cm.pl; pix < p1; pix += cm.PL) {
        const int64_t r = (int64_t)f * Npix + pix;
        const float* row = x + r * ldx;
        half_t* yrow = y + r * ldy;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int cv = cm.cv0 + j * cm.CVT;
            if (cv < cm.CV) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + cv * 4);
                half4v h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float o = fmaf(v[e] - mu[j][e], ga[j][e], be[j][e]);
                    if (silu) o = silu_f(o);
                    h[e] = (half_t)o;
                }
                *reinterpret_cast<half4v*>(yrow + cv * 4) = h;
            }
        }
    }
}

// ---- temporal GroupNorm + SiLU: one (b, pixel) = T rows of C; thread = channel pair --------------
// Block handles PB pixels; work items (pixel_local, channel pair), strided over 256 threads.
template <int T>
__global__ __launch_bounds__(256) void gn_temporal_kernel(const float* __restrict__ x, int B, int Npix, int C,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          half_t* __restrict__ y, int PB) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [PB][C/2][2] sums, then [PB][32][2] stats
    const int CP = C >> 1, cpg2 = (C / GROUPS) >> 1;
    float* s_part = sm;                           // PB*CP*2
    float* s_stat = sm + (size_t)PB * CP * 2;     // PB*32*2
    const int tid = threadIdx.x;
    const int64_t bp0 = (int64_t)blockIdx.x * PB;   // first (b*Npix + pixel) of the block
    const int64_t total = (int64_t)B * Npix;
    const int nwork = PB * CP;
    for (int wi = tid; wi < nwork; wi += 256) {
        const int pl = wi / CP, cp = wi - pl * CP;
        const int64_t bp = bp0 + pl;
        float s = 0.0f, q = 0.0f;
        if (bp < total) {
            const int64_t b = bp / Npix, pix = bp - b * Npix;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float2 v = *reinterpret_cast<const float2*>(x + ((b * T + t) * Npix + pix) * C + cp * 2);
                s += v.x + v.y; q = fmaf(v.x, v.x, q); q = fmaf(v.y, v.y, q);
            }
        }
        s_part[wi * 2] = s; s_part[wi * 2 + 1] = q;
    }
    __syncthreads();
    for (int gi = tid; gi < PB * GROUPS; gi += 256) {
        const int pl = gi / GROUPS, g = gi - pl * GROUPS;
        float s = 0.0f, q = 0.0f;
        for (int j = 0; j < cpg2; ++j) {
            const int wi = pl * CP + g * cpg2 + j;
            s += s_part[wi * 2]; q += s_part[wi * 2 + 1];
        }
        const float n = (float)(cpg2 * 2 * T);
        const float mean = s / n;
        const float var = fmaxf(q / n - mean * mean, 0.0f);
        s_stat[gi * 2] = mean; s_stat[gi * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int wi = tid; wi < nwork; wi += 256) {
        const int pl = wi / CP, cp = wi - pl * CP;
        const int64_t bp = bp0 + pl;
        if (bp >= total) continue;
        const int64_t b = bp / Npix, pix = bp - b * Npix;
        const int g = cp / cpg2;
        const float mean = s_stat[(pl * GROUPS + g) * 2], rstd = s_stat[(pl * GROUPS + g) * 2 + 1];
        const float g0 = gamma[cp * 2] * rstd, g1 = gamma[cp * 2 + 1] * rstd;
        const float b0 = beta[cp * 2], b1 = beta[cp * 2 + 1];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t off = ((b * T + t) * Npix + pix) * C + cp * 2;
            const float2 v = *reinterpret_cast<const float2*>(x + off);
            half2v h;
            h[0] = (half_t)silu_f(fmaf(v.x - mean, g0, b0));
            h[1] = (half_t)silu_f(fmaf(v.y - mean, g1, b1));
            *reinterpret_cast<half2v*>(y + off) = h;
        }
    }
}

// ---- LayerNorm: one wave per row, float4 vectors, two-pass variance in registers ------------------
constexpr int LN_MAXV = 12;    // C <= 4*64*12 = 3072
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, int M, int C,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        half_t* __restrict__ y, int ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int CV = C >> 2;
    const float* xr = x + row * ldx;
    f32x4 v[LN_MAXV];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int cv = lane + j * 64;
        if (cv < CV) {
            v[j] = *reinterpret_cast<const f32x4*>(xr + cv * 4);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int cv = lane + j * 64;
        if (cv < CV) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q = fmaf(d, d, q); }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    half_t* yr = y + row * ldy;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int cv = lane + j * 64;
        if (cv < CV) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + cv * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + cv * 4);
            half4v h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)fmaf((v[j][e] - mean) * rstd, g[e], b[e]);
            *reinterpret_cast<half4v*>(yr + cv * 4) = h;
        }
    }
}

}  // namespace

extern "C" int pnc_groupnorm_stats(const float* x, int ldx, int F, int Npix, int C,
                                   int pix_per_chunk, float* partial, void* stream) {
    if (!x || !partial || F < 1 || Npix < 1 || pix_per_chunk < 1) return PNC_EINVAL;
    if (C % 64 || C > 4 * 256 * MAXV || ldx % 4) return PNC_EINVAL;
    if ((uintptr_t)x & 15) return PNC_EALIGN;
    const int nchunk = (Npix + pix_per_chunk - 1) / pix_per_chunk;
    const int CV = C / 4, CVT = CV < 256 ? CV : 256, PL = 256 / CVT;
    const size_t lds = (size_t)2 * PL * C * sizeof(float);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunk, F), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       x, ldx, Npix, C, pix_per_chunk, partial);
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_apply(const float* x, int ldx, int F, int Npix, int C,
                                   int pix_per_chunk, const float* partial,
                                   const float* gamma, const float* beta, float eps, int silu,
                                   void* y16, int ldy, void* stream) {
    if (!x || !partial || !gamma || !beta || !y16 || F < 1 || Npix < 1 || pix_per_chunk < 1) return PNC_EINVAL;
    if (C % 64 || C > 4 * 256 * MAXV || ldx % 4 || ldy % 4) return PNC_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)y16 & 7)) return PNC_EALIGN;
    const int nchunk = (Npix + pix_per_chunk - 1) / pix_per_chunk;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(nchunk, F), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, ldx, Npix, C, pix_per_chunk, partial, gamma, beta, eps, silu,
                       reinterpret_cast<half_t*>(y16), ldy);
    return pnc_launch_status();
}

extern "C" int pnc_groupnorm_temporal_silu(const float* x, int B, int T, int Npix, int C,
                                           const float* gamma, const float* beta, float eps,
                                           void* y16, void* stream) {
    if (!x || !gamma || !beta || !y16 || B < 1 || Npix < 1) return PNC_EINVAL;
    if (C % 64 || T < 1 || T > 8) return PNC_EINVAL;
    const int CP = C / 2;
    int PB = 1024 / CP; if (PB < 1) PB = 1; if (PB > 16) PB = 16;
    const int64_t total = (int64_t)B * Npix;
    const unsigned blocks = (unsigned)((total + PB - 1) / PB);
    const size_t lds = ((size_t)PB * CP * 2 + (size_t)PB * GROUPS * 2) * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    half_t* y = reinterpret_cast<half_t*>(y16);
#define PNC_GNT(TT) case TT: hipLaunchKernelGGL(gn_temporal_kernel<TT>, dim3(blocks), dim3(256), lds, st, \
                                               x, B, Npix, C, gamma, beta, eps, y, PB); break;
    switch (T) {
        PNC_GNT(1) PNC_GNT(2) PNC_GNT(3) PNC_GNT(4) PNC_GNT(5) PNC_GNT(6) PNC_GNT(7) PNC_GNT(8)
    }
#undef PNC_GNT
    return pnc_launch_status();
}

extern "C" int pnc_layernorm(const float* x, int ldx, int M, int C,
                             const float* gamma, const float* beta, float eps,
                             void* y16, int ldy, void* stream) {
    if (!x || !gamma || !beta || !y16 || M < 1) return PNC_EINVAL;
    if (C % 4 || C > 4 * 64 * LN_MAXV || ldx % 4 || ldy % 4) return PNC_EINVAL;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) return PNC_EALIGN;
    if ((uintptr_t)y16 & 7) return PNC_EALIGN;
    const unsigned blocks = (unsigned)(((int64_t)M + 3) / 4);
    hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, ldx, M, C, gamma, beta, eps, reinterpret_cast<half_t*>(y16), ldy);
    return pnc_launch_status();
}
