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
// Operands are staged global -> registers -> LDS (16-B chunks, XOR-swizzled 128-B rows, two LDS
// stages, one barrier per K tile: the loads of tile t+1 are in flight while tile t feeds the MFMAs).
// Workgroup ids are remapped so that each XCD walks a contiguous range of tiles (W stays in its L2).
#include "common.h"

namespace {

constexpr int BK = 64;          // fp16 elements per K tile = 128 B per LDS row
constexpr int NTHREADS = 256;

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

// 16-byte chunk of 8 consecutive k starting at kc (kc % 8 == 0) for row state s; zeros outside.
template <int AMODE>
__device__ __forceinline__ half8v load_a_chunk(const PncGemmParams& p, const half_t* __restrict__ A,
                                               const RowState& s, int kc) {
    half8v z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!s.valid || kc >= p.K) return z;
    if (AMODE == PNC_A_PLAIN) {
        return *reinterpret_cast<const half8v*>(A + s.base + kc);
    } else if (AMODE == PNC_A_CONV3X3) {
        const int tap = kc / p.Cin, ci = kc - tap * p.Cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        int iy, ix; bool ok;
        if (p.upsample) {
            const int uy = s.y + ky - 1, ux = s.x + kx - 1;
            ok = (uy >= 0) && (uy < p.Hout) && (ux >= 0) && (ux < p.Wout);
            iy = uy >> 1; ix = ux >> 1;
        } else {
            iy = s.y * p.stride + ky - 1; ix = s.x * p.stride + kx - 1;
            ok = (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win);
        }
        if (!ok) return z;
        return *reinterpret_cast<const half8v*>(A + s.base + ((int64_t)iy * p.Win + ix) * p.Cin + ci);
    } else {
        const int tap = kc / p.Cin, ci = kc - tap * p.Cin;
        const int tt = s.y + tap - 1;
        if (tt < 0 || tt >= p.T) return z;
        return *reinterpret_cast<const half8v*>(A + s.base + (int64_t)(tap - 1) * p.Npix * p.Cin + ci);
    }
}

template <int AMODE, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const PncGemmParams p) {
    constexpr int MI = BM / WGM / 32;       // 32-row blocks per wave
    constexpr int NI = BN / WGN / 32;
    constexpr int A_IT = BM * 8 / NTHREADS; // 16-B chunks per thread per tile
    constexpr int B_IT = (BN * 8 + NTHREADS - 1) / NTHREADS;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // staging assignment: chunk column c (0..7) and rows r0 + 32*i
    const int sc = tid & 7, sr = tid >> 3;
    RowState rows[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) rows[i] = make_row<AMODE>(p, m0 + sr + 32 * i);

    half8v ra[A_IT], rb[B_IT];
    auto load_tile = [&](int kt) {
        const int kc = kt * BK + sc * 8;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = load_a_chunk<AMODE>(p, A, rows[i], kc);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int rr = sr + 32 * i;
            const int n = n0 + rr;
            half8v z = {0, 0, 0, 0, 0, 0, 0, 0};
            rb[i] = (rr < BN && n < p.N && kc < p.K)
                        ? *reinterpret_cast<const half8v*>(Wt + (int64_t)n * p.K + kc) : z;
        }
    };
    auto store_tile = [&](int stage) {
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            *reinterpret_cast<half8v*>(sa + lds_off128(sr + 32 * i, sc)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int rr = sr + 32 * i;
            if (rr < BN) *reinterpret_cast<half8v*>(sb + lds_off128(rr, sc)) = rb[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int ntiles = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frow = lane & 31, fk = lane >> 5;
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = (kt + 1) < ntiles;
        if (more) load_tile(kt + 1);
        const char* sa = smem + (kt & 1) * STAGE;
        const char* sb = sa + A_BYTES;
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
        }
        if (more) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ------------------------------ epilogue ------------------------------
    const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
    const int col = lane & 31;
    half_t* out16 = reinterpret_cast<half_t*>(p.out16);
    half_t* out16t = reinterpret_cast<half_t*>(p.out16t);
    const bool transposed = (out16t != nullptr) && (n0 >= p.n_split);

    if (p.geglu) {
        if (NI >= 2) {
            const int Nout = p.N >> 1;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int jp = 0; jp < NI / 2; ++jp) {
                    const int nv = nw + (2 * jp) * 32 + col, ng = nv + 32;
                    if (ng >= p.N) continue;
                    const float bv = p.bias ? p.bias[nv] : 0.0f, bg = p.bias ? p.bias[ng] : 0.0f;
                    const int on = (nw >> 1) + jp * 32 + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mw + i * 32 + mfma32_row(r, lane);
                        if (m >= p.M) continue;
                        const float v = (acc[i][2 * jp][r] + bv) * gelu_erf_f(acc[i][2 * jp + 1][r] + bg);
                        if (out16) out16[(int64_t)m * p.ldc16 + on] = (half_t)v;
                        if (p.out32) p.out32[(int64_t)m * p.ldc32 + on] = v;
                    }
                }
            }
            (void)Nout;
        }
        return;
    }

#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = nw + j * 32 + col;
            const bool nok = n < p.N;
            const float bn = (p.bias && nok) ? p.bias[n] : 0.0f;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float v[4];
                const int mb = mw + i * 32 + 8 * r4 + 4 * (lane >> 5);   // rows mb..mb+3
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = mb + q;
                    float x = acc[i][j][r4 * 4 + q] + bn;
                    if (nok && m < p.M) {
                        if (p.rowbias) x += p.rowbias[(int64_t)((m / p.rb_rows) % p.rb_mod) * p.N + n];
                        if (p.act == PNC_ACT_SILU) x = silu_f(x);
                        if (p.res1) x += p.res1[(int64_t)m * p.ldr1 + n];
                        if (p.res2) x += p.res2[(int64_t)m * p.ldr2 + n];
                    }
                    v[q] = x;
                }
                if (!nok) continue;
                if (transposed) {
                    const int g = mb / p.t_rows, tr = mb - g * p.t_rows;
                    half_t* dst = out16t + (int64_t)g * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + tr;
                    const bool vec = (mb + 3 < p.M) && (tr + 3 < p.t_rows) && ((p.ldt & 3) == 0) &&
                                     ((p.t_gstride & 3) == 0) && ((tr & 3) == 0);
                    if (vec) {
                        half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *reinterpret_cast<half4v*>(dst) = h;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int m = mb + q;
                            if (m < p.M) {
                                const int g2 = m / p.t_rows, t2 = m - g2 * p.t_rows;
                                out16t[(int64_t)g2 * p.t_gstride + (int64_t)(n - p.n_split) * p.ldt + t2] = (half_t)v[q];
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = mb + q;
                        if (m >= p.M) continue;
                        if (p.out32) p.out32[(int64_t)m * p.ldc32 + n] = v[q];
                        if (out16) out16[(int64_t)m * p.ldc16 + n] = (half_t)v[q];
                    }
                }
            }
        }
    }
}

template <int AMODE, int BM, int BN, int WGM, int WGN>
int launch(const PncGemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * 128;
    static bool attr_done = false;   // per-instantiation; idempotent
    auto kern = gemm_kernel<AMODE, BM, BN, WGM, WGN>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHREADS), lds, st, p);
    return pnc_launch_status();
}

template <int AMODE>
int dispatch(const PncGemmParams& p, hipStream_t st) {
    if (p.N <= 32 && !p.geglu) return launch<AMODE, 128, 32, 4, 1>(p, st);
    return launch<AMODE, 128, 128, 2, 2>(p, st);
}

}  // namespace

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
