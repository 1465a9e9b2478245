// misc.hip — small HBM-bound helpers of the denoising path: time-embedding MLP (small-M fp32
// linear), sinusoidal embedding, NCHW <-> channels-last conversion, skip/control concat-add.
#include "common.h"
#include <atomic>

namespace {

constexpr int SM_MAXM = 16;

// one wave per output column n: out[m][n] = sum_k f(a[m][k]) W[n][k] + bias[n]
__global__ __launch_bounds__(256) void linear_smallm_kernel(const float* __restrict__ a, int lda,
                                                            const half_t* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ out, int ldo, int M, int N,
                                                            int K, int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[SM_MAXM];
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) acc[m] = 0.0f;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
        const half8v w = *reinterpret_cast<const half8v*>(W + (int64_t)n * K + k0);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = (float)w[e];
#pragma unroll
        for (int m = 0; m < SM_MAXM; ++m) {
            if (m < M) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(a + (int64_t)m * lda + k0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(a + (int64_t)m * lda + k0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u0 = silu_in ? silu_f(x0[e]) : x0[e];
                    const float u1 = silu_in ? silu_f(x1[e]) : x1[e];
                    acc[m] = fmaf(u0, wf[e], acc[m]);
                    acc[m] = fmaf(u1, wf[e + 4], acc[m]);
                }
            }
        }
    }
    const float b = bias ? bias[n] : 0.0f;
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) {
        if (m < M) {
            float v = wave_sum(acc[m]) + b;
            if (silu_out) v = silu_f(v);
            if (lane == 0) out[(int64_t)m * ldo + n] = v;
        }
    }
}

// The same linear for MANY sites in one launch (round 5): W holds the sites' weight rows back to back (N = sum of the sites'
// widths), the output is laid out one contiguous [Mtot x width] block per site (segment), blocks back to back — what each site's
// consumer (the row-bias operand of its GEMM epilogue: [rows][N_site]) reads.  One wave per SEG_NC consecutive columns: the 16 x K
// activation rows are read once per wave instead of once per column.  Per column the K order, the fma chain and the wave
// reduction are those of linear_smallm_kernel: bit-identical to one launch per site.
constexpr int SEG_NC = 4;
struct SmallmSegs {
    int nseg;
    int start[PNC_SMALLM_MAX_SEGS + 1];
};

__global__ __launch_bounds__(256) void linear_smallm_seg_kernel(const float* __restrict__ a, int lda,
                                                                const half_t* __restrict__ W,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ out, int M, int m0, int Mtot, int N,
                                                                int K, int silu_in, int silu_out, SmallmSegs segs) {
    const int lane = threadIdx.x & 63;
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SEG_NC;
    if (n0 >= N) return;
    float acc[SEG_NC][SM_MAXM];
#pragma unroll
    for (int c = 0; c < SEG_NC; ++c)
#pragma unroll
        for (int m = 0; m < SM_MAXM; ++m) acc[c][m] = 0.0f;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
        float wf[SEG_NC][8];
#pragma unroll
        for (int c = 0; c < SEG_NC; ++c) {
            const half8v w = *reinterpret_cast<const half8v*>(W + (int64_t)(n0 + c) * K + k0);
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[c][e] = (float)w[e];
        }
#pragma unroll
        for (int m = 0; m < SM_MAXM; ++m) {
            if (m < M) {
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(a + (int64_t)m * lda + k0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(a + (int64_t)m * lda + k0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u0 = silu_in ? silu_f(x0[e]) : x0[e];
                    const float u1 = silu_in ? silu_f(x1[e]) : x1[e];
#pragma unroll
                    for (int c = 0; c < SEG_NC; ++c) {
                        acc[c][m] = fmaf(u0, wf[c][e], acc[c][m]);
                        acc[c][m] = fmaf(u1, wf[c][e + 4], acc[c][m]);
                    }
                }
            }
        }
    }
    int s = 0;                                              // wave-uniform: a wave's columns lie in one segment (widths % SEG_NC == 0)
    while (s + 1 < segs.nseg && n0 >= segs.start[s + 1]) ++s;
    const int s0 = segs.start[s], len = segs.start[s + 1] - s0;
    float* o = out + (int64_t)s0 * Mtot + (int64_t)m0 * len + (n0 - s0);
#pragma unroll
    for (int c = 0; c < SEG_NC; ++c) {
        const float b = bias ? bias[n0 + c] : 0.0f;
#pragma unroll
        for (int m = 0; m < SM_MAXM; ++m) {
            if (m < M) {
                float v = wave_sum(acc[c][m]) + b;
                if (silu_out) v = silu_f(v);
                if (lane == 0) o[(int64_t)m * len + c] = v;
            }
        }
    }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, int F, int dim,
                                          const float* __restrict__ freqs, float* __restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * half) return;
    const int f = i / half, j = i - f * half;
    // freqs[j] = exp(-ln(max_period) * j / half) is tabulated by the host exactly as
    // diffusionmodules/util.py:236-241 does (fp32), so only cos/sin run here.
    const float arg = (float)t[f] * freqs[j];
    out[(int64_t)f * dim + j] = cosf(arg);
    out[(int64_t)f * dim + half + j] = sinf(arg);
    if ((dim & 1) && j == 0) out[(int64_t)f * dim + dim - 1] = 0.0f;
}

// thread per (f, pixel): gather channels (stride Npix), write Cpad fp16 contiguous
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const float* __restrict__ a, int C1,
                                                             const float* __restrict__ a_scale, int a_frames,
                                                             const float* __restrict__ b, int C2, int F,
                                                             int Npix, int Cpad, half_t* __restrict__ out,
                                                             half_t* __restrict__ out_lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)F * Npix) return;
    const int64_t f = i / Npix, pix = i - f * Npix;
    const int64_t fa = f % a_frames;                 // CFG batch doubling: both halves read the same latent
    const float sa = a_scale ? a_scale[f] : 1.0f;
    half_t* o = out + i * Cpad;
    for (int c0 = 0; c0 < Cpad; c0 += 8) {
        half8v h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float v = 0.0f;
            if (c < C1) v = a[(fa * C1 + c) * Npix + pix] * sa;
            else if (c < C1 + C2) v = b[(f * C2 + (c - C1)) * Npix + pix];
            h[e] = (half_t)v;
            l[e] = (half_t)((v - (float)h[e]) * PNC_LO_SCALE);
        }
        *reinterpret_cast<half8v*>(o + c0) = h;
        if (out_lo) *reinterpret_cast<half8v*>(out_lo + i * Cpad + c0) = l;
    }
}

__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ x, int ld, int F,
                                                             int Npix, int C, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)F * Npix) return;
    const int64_t f = i / Npix, pix = i - f * Npix;
    for (int c = 0; c < C; ++c) out[(f * C + c) * Npix + pix] = x[i * ld + c];
}

// float4 granularity over the concatenated row
__global__ __launch_bounds__(256) void concat_add_kernel(const float* __restrict__ a, int C1,
                                                         const float* __restrict__ s,
                                                         const float* __restrict__ c, int C2, int64_t M,
                                                         float* __restrict__ out32, half_t* __restrict__ out16,
                                                         void* __restrict__ out16_lo, int lo_fmt) {
    const int CT = C1 + C2, V = CT >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * V) return;
    const int64_t m = i / V;
    const int ch = (int)(i - m * V) * 4;
    f32x4 v;
    if (ch < C1) {
        v = *reinterpret_cast<const f32x4*>(a + m * C1 + ch);
    } else {
        const int c2 = ch - C1;
        v = *reinterpret_cast<const f32x4*>(s + m * C2 + c2);
        if (c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(c + m * C2 + c2);
            v += w;
        }
    }
    if (out32) *reinterpret_cast<f32x4*>(out32 + m * CT + ch) = v;
    if (out16) {
        half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4v*>(out16 + m * CT + ch) = h;
        if (out16_lo) {
            const float o[4] = {v[0], v[1], v[2], v[3]};
            store_lo4(out16_lo, lo_fmt, m * CT + ch, o, h);
        }
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                  int64_t n4, float* __restrict__ y32, half_t* __restrict__ y16,
                                                  void* __restrict__ y16_lo, int lo_fmt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    if (a) { const f32x4 w = *reinterpret_cast<const f32x4*>(a + i * 4); v += w; }
    if (y32) *reinterpret_cast<f32x4*>(y32 + i * 4) = v;
    if (y16) {
        half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4v*>(y16 + i * 4) = h;
        if (y16_lo) {
            const float o[4] = {v[0], v[1], v[2], v[3]};
            store_lo4(y16_lo, lo_fmt, i * 4, o, h);
        }
    }
}

// Exit of a sampler step: eps tokens of the CFG batch -> next latent, one pass.  The arithmetic
// follows the reference's sequence of fp32 roundings (no contraction into FMAs), so a trajectory matches the reference's
// sampler to the last few ulps.
__global__ __launch_bounds__(256) void cfg_euler_step_kernel(const float* __restrict__ eps, int ld, int T, int Npix, int C,
                                                             int cfg, float scale, const float* __restrict__ x,
                                                             const float* __restrict__ c_out,
                                                             const float* __restrict__ sigma,
                                                             const float* __restrict__ sigma_next,
                                                             float* __restrict__ xn) {
    // every product and sum below is rounded on its own, like the reference's separate elementwise kernels.  The operators
    // are written out under `fp contract(off)` (HIP's __fmul_rn / __fadd_rn are plain inline operators compiled under the
    // default contract(fast), so they DO fuse into FMAs)
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)T * Npix) return;
    const int64_t t = i / Npix, pix = i - t * Npix;
    const float sg = sigma[t], dt = sigma_next[t] - sg, co = c_out[t];
    const float* eu = eps + i * ld;                               // uncond half first (guiders.py:36)
    const float* ec = eps + ((int64_t)(cfg ? T : 0) * Npix + i) * ld;
    for (int c = 0; c < C; ++c) {
        const int64_t o = (t * C + c) * Npix + pix;
        const float xv = x[o];
        const float pc = ec[c] * co;
        const float dc = pc + xv;                                        // eps * c_out + x * c_skip   (c_skip = 1)
        float d = dc;
        if (cfg) {
            const float pu = eu[c] * co;
            const float du = pu + xv;
            const float g = scale * (dc - du);
            d = du + g;                                                  // x_u + scale * (x_c - x_u)
        }
        const float dir = (xv - d) / sg;                                 // (x - denoised) / sigma
        const float stp = dt * dir;
        xn[o] = xv + stp;                                                // x + (sigma_next - sigma) * d
    }
}

}  // namespace

extern "C" int pnc_cfg_euler_step(const float* eps_tok, int ld, int T, int Npix, int C, int cfg, float scale,
                                  const float* x, const float* c_out, const float* sigma, const float* sigma_next,
                                  float* x_next, void* stream) {
    if (!eps_tok || !x || !c_out || !sigma || !sigma_next || !x_next || T < 1 || Npix < 1 || C < 1 || ld < C) return PNC_EINVAL;
    const int64_t n = (int64_t)T * Npix;
    hipLaunchKernelGGL(cfg_euler_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), eps_tok, ld, T, Npix, C, cfg, scale, x, c_out, sigma, sigma_next, x_next);
    return pnc_launch_status();
}

extern "C" const char* pnc_version(void) { return "panacea_hip 0.4.0 gfx950"; }
// the digest of the sources this object was compiled from (panacea_amd/build.py passes it to this translation unit)
#ifndef PNC_BUILD_DIGEST
#define PNC_BUILD_DIGEST "unstamped"
#endif
// (stored behind a marker so that build.py can read it from the FILE: dlopen-ing a library to ask it would pin the old image in
// the asking process and make a rebuild + load in the same process see the stale one)
static const char k_build_digest[] = "pnc-build-digest:" PNC_BUILD_DIGEST;
extern "C" const char* pnc_build_digest(void) { return k_build_digest + 17; }

static std::atomic<int> g_options[PNC_OPT_COUNT] = {{1}, {0}, {0}, {1}, {1}, {0}, {1}, {3}, {8}, {1}, {4}, {12}};

int pnc_get_option(int option) { return g_options[option].load(std::memory_order_relaxed); }

extern "C" int pnc_set_option(int option, int value) {
    if (option < 0 || option >= PNC_OPT_COUNT) return PNC_EINVAL;
    return g_options[option].exchange(value, std::memory_order_relaxed);
}

extern "C" int pnc_linear_smallm(const float* a, int lda, const void* W, const float* bias,
                                 float* out, int ldo, int M, int N, int K, int silu_in, int silu_out,
                                 void* stream) {
    if (!a || !W || !out || M < 1 || M > SM_MAXM || N < 1 || K < 8) return PNC_EINVAL;
    if (K % 8 || lda % 4) return PNC_EINVAL;
    if (((uintptr_t)a | (uintptr_t)W) & 15) return PNC_EALIGN;
    hipLaunchKernelGGL(linear_smallm_kernel, dim3((N + 3) / 4), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, lda, reinterpret_cast<const half_t*>(W), bias,
                       out, ldo, M, N, K, silu_in, silu_out);
    return pnc_launch_status();
}

extern "C" int pnc_linear_smallm_segments(const float* a, int lda, const void* W, const float* bias, float* out, int M,
                                          int m0, int Mtot, int N, int K, const int32_t* seg_start, int nseg, int silu_in,
                                          int silu_out, void* stream) {
    if (!a || !W || !out || !seg_start || M < 1 || M > SM_MAXM || m0 < 0 || m0 + M > Mtot || N < 1 || K < 8) return PNC_EINVAL;
    if (K % 8 || lda % 4 || nseg < 1 || nseg > PNC_SMALLM_MAX_SEGS) return PNC_EINVAL;
    if (((uintptr_t)a | (uintptr_t)W) & 15) return PNC_EALIGN;
    SmallmSegs segs;
    segs.nseg = nseg;
    if (seg_start[0] != 0 || seg_start[nseg] != N) return PNC_EINVAL;
    for (int i = 0; i <= nseg; ++i) {
        if (seg_start[i] % SEG_NC || (i > 0 && seg_start[i] <= seg_start[i - 1])) return PNC_EINVAL;
        segs.start[i] = seg_start[i];
    }
    const int waves = N / SEG_NC;
    hipLaunchKernelGGL(linear_smallm_seg_kernel, dim3((waves + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, lda,
                       reinterpret_cast<const half_t*>(W), bias, out, M, m0, Mtot, N, K, silu_in, silu_out, segs);
    return pnc_launch_status();
}

extern "C" int pnc_timestep_embedding(const int64_t* t, int F, int dim, const float* freqs,
                                      float* out, void* stream) {
    if (!t || !out || !freqs || F < 1 || dim < 2) return PNC_EINVAL;
    const int n = F * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), t, F, dim, freqs, out);
    return pnc_launch_status();
}

extern "C" int pnc_nchw_to_tokens_f16(const float* a, int C1, const float* a_scale, int a_frames, const float* b, int C2,
                                      int F, int Npix, int Cpad, void* out16, void* out16_lo, void* stream) {
    if (!a || !out16 || F < 1 || Npix < 1 || C1 < 1 || C2 < 0 || (C2 > 0 && !b)) return PNC_EINVAL;
    if (a_frames < 1 || a_frames > F) return PNC_EINVAL;
    if (Cpad % 8 || Cpad < C1 + C2) return PNC_EINVAL;
    if (((uintptr_t)out16 | (uintptr_t)out16_lo) & 15) return PNC_EALIGN;
    const int64_t n = (int64_t)F * Npix;
    hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, C1, a_scale, a_frames, b, C2, F, Npix, Cpad,
                       reinterpret_cast<half_t*>(out16), reinterpret_cast<half_t*>(out16_lo));
    return pnc_launch_status();
}

extern "C" int pnc_tokens_to_nchw_f32(const float* x, int ld, int F, int Npix, int C,
                                      float* out, void* stream) {
    if (!x || !out || F < 1 || Npix < 1 || C < 1 || ld < C) return PNC_EINVAL;
    const int64_t n = (int64_t)F * Npix;
    hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, ld, F, Npix, C, out);
    return pnc_launch_status();
}

extern "C" int pnc_concat_add(const float* a, int C1, const float* s, const float* c, int C2,
                              int64_t M, float* out32, void* out16, void* out16_lo, int lo_fmt, void* stream) {
    if (!a || !s || M < 1 || C1 % 4 || C2 % 4 || C1 < 4 || C2 < 4) return PNC_EINVAL;
    if (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3) return PNC_EINVAL;
    if ((!out32 && !out16) || (out16_lo && !out16)) return PNC_EINVAL;
    const int64_t n = M * ((C1 + C2) / 4);
    hipLaunchKernelGGL(concat_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, C1, s, c, C2, M, out32,
                       reinterpret_cast<half_t*>(out16), out16_lo, lo_fmt);
    return pnc_launch_status();
}

extern "C" int pnc_add_f32(const float* x, const float* a, int64_t n, float* y32, void* y16, void* y16_lo, int lo_fmt,
                           void* stream) {
    if (!x || n < 4 || n % 4 || (!y32 && !y16) || (y16_lo && !y16)) return PNC_EINVAL;
    if (lo_fmt != PNC_LO_F16 && lo_fmt != PNC_LO_E4M3) return PNC_EINVAL;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, a, n4, y32, reinterpret_cast<half_t*>(y16),
                       y16_lo, lo_fmt);
    return pnc_launch_status();
}

// row softmax: one 256-thread block per row, the row lives in registers (<= 16 x float4 per thread), one HBM read
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int64_t lds, int N, float scale,
                                                           int causal, int n_valid,
                                                           half_t* __restrict__ p, int64_t ldp) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = s + (int64_t)blockIdx.x * lds;
    half_t* prow = p + (int64_t)blockIdx.x * ldp;
    const int nv = N >> 2;
    // keys this row may see: the first n_valid columns, and with a causal mask only those up to the row's own index
    const int lim = causal ? min(n_valid, (int)blockIdx.x + 1) : n_valid;
    f32x4 v[16];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = tid + j * 256;
        if (c < nv) {
            v[j] = *reinterpret_cast<const f32x4*>(row + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = (c * 4 + e < lim) ? v[j][e] : -3.0e38f;      // masked: exp2 -> 0
            m = fmaxf(fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])), m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = scale * 1.44269504088896340736f;
    const float off = m * sc;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = tid + j * 256;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[j][e] = __builtin_amdgcn_exp2f(fmaf(v[j][e], sc, -off)); sum += v[j][e]; }
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (((red[4] + red[5]) + red[6]) + red[7]);     // fixed order: deterministic
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = tid + j * 256;
        if (c < nv) {
            half4v h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)(v[j][e] * inv);
            *reinterpret_cast<half4v*>(prow + c * 4) = h;
        }
    }
}

extern "C" int pnc_softmax_rows_f16(const float* s, int64_t lds, int M, int N, float scale, int causal, int n_valid,
                                    void* p16, int64_t ldp, void* stream) {
    if (!s || !p16 || M < 1 || N < 4 || N > 16384 || (N & 3) || n_valid > N) return PNC_EINVAL;
    if (n_valid <= 0) n_valid = N;
    if ((lds & 3) || (ldp & 3) || lds < N || ldp < N) return PNC_EINVAL;
    if (((uintptr_t)s & 15) || ((uintptr_t)p16 & 7)) return PNC_EALIGN;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)M), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), s, lds, N,
                       scale, causal, n_valid, reinterpret_cast<half_t*>(p16), ldp);
    return pnc_launch_status();
}

extern "C" int pnc_cast_f16(const float* x, int64_t n, void* y16, void* y16_lo, int lo_fmt, void* stream) {
    return pnc_add_f32(x, nullptr, n, nullptr, y16, y16_lo, lo_fmt, stream);
}

PNC_DEFINE_TU_COLLECT(misc)
void pnc_tu_collect_gemm(unsigned int*, hipStream_t);
void pnc_tu_collect_gemm_plain(unsigned int*, hipStream_t);
void pnc_tu_collect_gemm_conv3x3(unsigned int*, hipStream_t);
void pnc_tu_collect_gemm_conv1d(unsigned int*, hipStream_t);
void pnc_tu_collect_gemm_stencil_tile(unsigned int*, hipStream_t);
void pnc_tu_collect_norm(unsigned int*, hipStream_t);

extern "C" int pnc_range_monitor_collect(unsigned int* out, void* stream) {
    if (!out) return PNC_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    pnc_tu_collect_gemm(out, st);
    pnc_tu_collect_gemm_plain(out, st);
    pnc_tu_collect_gemm_conv3x3(out, st);
    pnc_tu_collect_gemm_conv1d(out, st);
    pnc_tu_collect_gemm_stencil_tile(out, st);
    pnc_tu_collect_norm(out, st);
    pnc_tu_collect_misc(out, st);
    return pnc_launch_status();
}
