// gemm.hip — C-ABI entry of the MFMA GEMM family: argument validation, epilogue-variant selection, split-K reduce
// kernel, GEGLU Phi table.  The kernels themselves are instantiated per A-gather mode in gemm_plain.hip /
// gemm_conv3x3.hip / gemm_conv1d.hip from the template in gemm_kernel.h.
#include "gemm_kernel.h"
#include <math.h>
#include <mutex>

namespace pnc_gemm {

int dispatch_plain(const PncGemmParams& p, unsigned epi, hipStream_t st, bool* ln_fused);
int dispatch_conv3x3(const PncGemmParams& p, unsigned epi, hipStream_t st);
int dispatch_conv1d(const PncGemmParams& p, unsigned epi, hipStream_t st);

__device__ __attribute__((aligned(16))) float g_phi_table[2 * PHI_N];

// One-time upload of the Phi table per device (host-computed in double).  Synchronous, hence not legal during stream
// capture: the first GEGLU GEMM of a process has to run eagerly (every warm-up does).  Thread-safe.
const float* phi_table_device(hipStream_t st, int* rc) {
    static std::mutex mu;
    static const float* dev_ptr[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    const float*& ptr = dev_ptr[dev & 63];
    *rc = PNC_OK;
    if (ptr) return ptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { *rc = PNC_EINVAL; return nullptr; }
    static float host[2 * PHI_N];
    auto phi = [](double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)); };
    for (int i = 0; i < PHI_N; ++i) {
        const double x0 = (double)PHI_X0 + (double)i / PHI_SCALE, x1 = (double)PHI_X0 + (double)(i + 1) / PHI_SCALE;
        host[2 * i] = (float)phi(x0);
        host[2 * i + 1] = (float)(phi(x1) - phi(x0));
    }
    void* d = nullptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phi_table), host, sizeof(host)) != hipSuccess ||
        hipGetSymbolAddress(&d, HIP_SYMBOL(g_phi_table)) != hipSuccess) {
        *rc = (int)hipGetLastError();
        return nullptr;
    }
    ptr = reinterpret_cast<const float*>(d);
    return ptr;
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
    f32x4 s0 = ld4(src), s1 = ld4(src + 4);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 t0 = ld4(src + s * mn), t1 = ld4(src + s * mn + 4);
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
    if (p.act == PNC_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (p.act == PNC_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gelu_erf_call(v[e]);
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
        half_t* ol = reinterpret_cast<half_t*>(p.out16_lo);
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half_t h = (half_t)v[e];
            op[e] = h;
            r[e] = (v[e] - (float)h) * LO_SCALE;
        }
        if (ol && p.out_lo_fmt == PNC_LO_E4M3) {
            unsigned char* o8 = reinterpret_cast<unsigned char*>(p.out16_lo) + (int64_t)m * p.ldc16 + ncol;
            const unsigned w0 = pack4_e4m3(r[0], r[1], r[2], r[3]), w1 = pack4_e4m3(r[4], r[5], r[6], r[7]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o8[e] = (unsigned char)(w0 >> (8 * e)); o8[4 + e] = (unsigned char)(w1 >> (8 * e)); }
        } else if (ol) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ol[(int64_t)m * p.ldc16 + ncol + e] = (half_t)r[e];
        }
    }
}

int launch_splitk_reduce(const PncGemmParams& p, int ksplit, hipStream_t st) {
    const int64_t work = (int64_t)p.M * (p.N >> 3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, p, ksplit);
    return pnc_launch_status();
}

static inline bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

// Vector contract of the fast epilogues: every global access of the epilogue is a 16-byte vector on 8 consecutive
// columns.  Returns the EPI bit set of the launch, or E_GENERIC when the contract does not hold (or the option
// combination has no specialised variant).
static unsigned select_epilogue(const PncGemmParams& p) {
    if (p.geglu) return E_GEGLU | E_O16;          // validated separately: always inside the contract
    bool ok = (p.N % 8 == 0) && (!p.bias || al16(p.bias));
    if (p.out32) ok = ok && (p.ldc32 % 4 == 0) && al16(p.out32);
    if (p.out16) ok = ok && (p.ldc16 % 8 == 0) && al16(p.out16) && (!p.out16_lo || al16(p.out16_lo));
    if (p.res1) ok = ok && (p.ldr1 % 4 == 0) && al16(p.res1);
    if (p.res2) ok = ok && (p.ldr2 % 4 == 0) && al16(p.res2);
    if (p.rowbias) ok = ok && al16(p.rowbias);
    if (p.out16t) ok = ok && (p.M % 8 == 0) && (p.t_rows % 8 == 0) && (p.ldt % 8 == 0) && (p.t_gstride % 8 == 0) &&
                       al16(p.out16t) && !p.res1 && !p.res2 && !p.rowbias && !p.out32 && p.act == PNC_ACT_NONE &&
                       (p.n_split == 0 || p.out16);
    const int nstreams = (p.res1 != nullptr) + (p.res2 != nullptr) + (p.rowbias != nullptr);
    if (nstreams == 3 || (nstreams && p.act != PNC_ACT_NONE)) ok = false;
    if (p.act == PNC_ACT_GELU && (p.out32 || p.out16t || !p.out16)) ok = false;   // one specialised GELU variant: fp16 out
    if (!ok) return E_GENERIC;
    unsigned e = p.act == PNC_ACT_GELU ? E_GELU : 0;
    if (p.res1 || p.res2) e |= E_R1;              // a single residual is passed to the kernel as res1
    if (p.res1 && p.res2) e |= E_R2;
    if (p.rowbias) e |= E_RB;
    if (p.out32) e |= E_O32;
    if (p.out16) e |= E_O16;
    if (p.out16t) e |= E_VT | E_O16;
    return e;
}

}  // namespace pnc_gemm

using namespace pnc_gemm;

static int validate(const PncGemmParams& p) {
    if (p.struct_bytes != (int32_t)sizeof(PncGemmParams)) return PNC_EABI;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.W) return PNC_EINVAL;
    if (p.K % 8) return PNC_EINVAL;                       // 16-byte operand chunks
    if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.A_lo) & 15) return PNC_EALIGN;
    if (p.a_mode == PNC_A_PLAIN) {
        if (p.lda % 8 || p.lda < p.K) return PNC_EALIGN;
    } else if (p.a_mode == PNC_A_CONV3X3) {
        if (p.Cin % 8 || p.K != 9 * p.Cin || p.Hin <= 0 || p.Win <= 0) return PNC_EINVAL;
        if (p.stride != 1 && p.stride != 2) return PNC_EINVAL;
        if (p.upsample && (p.stride != 1 || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win)) return PNC_EINVAL;
        if (p.conv_pad_br && (p.upsample || p.stride != 2)) return PNC_EINVAL;
        if (p.M % (p.Hout * p.Wout) || p.Wout >= 65536 || p.M / (p.Hout * p.Wout) >= 32768) return PNC_EINVAL;
        if (p.x_halo_off < 0 || (p.x_halo_off && (p.conv_pad_br || (p.x_halo_off & 15)))) return PNC_EINVAL;
        if (p.x_halo_off + 2 * (int64_t)(p.M / (p.Hout * p.Wout)) * p.Hin * p.Cin >= (int64_t)1 << 30) return PNC_EINVAL;   // 2^31 bytes
    } else if (p.a_mode == PNC_A_CONV1D_T) {
        if (p.Cin % 8 || p.K != 3 * p.Cin || p.T <= 0 || p.Npix <= 0) return PNC_EINVAL;
        if (p.M % (p.T * p.Npix)) return PNC_EINVAL;
    } else {
        return PNC_EINVAL;
    }
    if (p.a_mode != PNC_A_CONV3X3 && p.x_halo_off) return PNC_EINVAL;
    if (p.gn_part && (p.a_mode != PNC_A_CONV1D_T || !p.out32 || (p.N % 64) || (p.ldc32 % 4) || p.out16t || p.geglu)) return PNC_EINVAL;
    if (p.act != PNC_ACT_NONE && p.act != PNC_ACT_SILU && p.act != PNC_ACT_GELU) return PNC_EINVAL;
    if (p.ldw != 0 && (p.ldw < p.K || p.ldw % 8)) return PNC_EALIGN;
    if (p.rowbias && (p.rb_rows <= 0 || p.rb_mod <= 0)) return PNC_EINVAL;
    if (p.geglu) {
        if ((p.N % 64) || p.out16t || p.out32 || p.res1 || p.res2 || p.rowbias || p.act != PNC_ACT_NONE || !p.out16)
            return PNC_EINVAL;
        if ((p.ldc16 % 8) || !al16(p.out16) || (p.out16_lo && !al16(p.out16_lo)) || (p.bias && !al16(p.bias)))
            return PNC_EALIGN;
    }
    if (p.out16t && ((p.n_split % 128) || p.t_rows <= 0 || p.N <= 32)) return PNC_EINVAL;
    if (p.out16_lo && !p.out16) return PNC_EINVAL;
    if ((p.a_lo_fmt != PNC_LO_F16 && p.a_lo_fmt != PNC_LO_E4M3) || (p.out_lo_fmt != PNC_LO_F16 && p.out_lo_fmt != PNC_LO_E4M3))
        return PNC_EINVAL;
    if (p.out16_lo && p.out_lo_fmt == PNC_LO_E4M3 && p.geglu) return PNC_EINVAL;        // (the GEGLU hidden is never a split operand)
    if (p.A_lo && p.a_lo_fmt == PNC_LO_E4M3) {
        // e4m3 lo pass: 16-byte chunks of 16 consecutive k on both operands, and the weight side of the pass
        if (!p.W_lo || p.w_lo_exp < 1 || p.w_lo_exp > 254) return PNC_EINVAL;
        if ((uintptr_t)p.W_lo & 15) return PNC_EALIGN;
        if (p.K % 16 || (p.ldw_lo != 0 && (p.ldw_lo < p.K || p.ldw_lo % 16))) return PNC_EALIGN;
        if (p.a_mode == PNC_A_PLAIN ? (p.lda % 16 != 0) : (p.Cin % 64 != 0)) return PNC_EALIGN;
    }
    if (!p.out32 && !p.out16 && !p.out16t) return PNC_EINVAL;
    if (p.ln_out16) {           // fused / trailing LayerNorm of the fp32 output rows (same limits as pnc_layernorm)
        if (!p.out32 || p.geglu || p.out16t || !p.ln_gamma || !p.ln_beta) return PNC_EINVAL;
        if (p.N % 4 || p.N > 4 * 64 * 12 || p.ldc32 % 4 || p.ldln % 4 || p.ldln < p.N) return PNC_EINVAL;
        if (!al16(p.out32) || !al16(p.ln_gamma) || !al16(p.ln_beta) || (((uintptr_t)p.ln_out16) & 7)) return PNC_EALIGN;
    }
    return PNC_OK;
}

extern "C" int pnc_abi_version(void) { return PNC_ABI_VERSION; }

extern "C" int64_t pnc_gemm_workspace_floats(const PncGemmParams* pp) {
    if (!pp || pp->struct_bytes != (int32_t)sizeof(PncGemmParams)) return 0;
    if (pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
    if (pp->N <= 32 && !pp->geglu) return 0;
    const int ks = splitk_slices(*pp);
    return ks > 1 ? (int64_t)ks * pp->M * pp->N : 0;
}

// the epilogue variant of a launch incl. the E_LN decision that does not depend on the tile (dispatch_plain settles the rest)
static unsigned epilogue_with_ln(const PncGemmParams& p) {
    unsigned epi = select_epilogue(p);
    // LayerNorm fused into the epilogue: plain A, fp32 output only, at most one added stream, 16-byte aligned fp16 rows
    if (p.ln_out16 && p.a_mode == PNC_A_PLAIN && (epi == E_O32 || epi == (E_RB | E_O32) || epi == (E_R1 | E_O32)) &&
        (p.ldln % 8 == 0) && al16(p.ln_out16) && (pnc_get_option(PNC_OPT_GEMM_FUSE_LN) & 1))
        epi |= E_LN;
    return epi;
}

static void normalise(PncGemmParams& p) {
    if (!p.out16t) p.n_split = p.N;
    if (p.ldw == 0) p.ldw = p.K;
    if (p.ldw_lo == 0) p.ldw_lo = p.K;
    if (!p.res1 && p.res2) { p.res1 = p.res2; p.ldr1 = p.ldr2; p.res2 = nullptr; }   // fp32 addition commutes bit-exactly for two terms
}

extern "C" int pnc_gemm_fuses_layernorm(const PncGemmParams* pp) {
    if (!pp || validate(*pp) != PNC_OK || !pp->ln_out16) return 0;
    PncGemmParams p = *pp;
    normalise(p);
    if (!(epilogue_with_ln(p) & E_LN)) return 0;
    const TileChoice tc = choose_tile(p);
    return ln_whole_rows(p, tc) ? 1 : 0;
}

extern "C" int pnc_gemm_f16(const PncGemmParams* pp, void* stream) {
    if (!pp) return PNC_EINVAL;
    const int rc = validate(*pp);
    if (rc != PNC_OK) return rc;
    PncGemmParams p = *pp;
    normalise(p);
    const unsigned epi = epilogue_with_ln(p);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool ln_fused = false;
    int rc2;
    switch (p.a_mode) {
        case PNC_A_PLAIN: rc2 = dispatch_plain(p, epi, st, &ln_fused); break;
        case PNC_A_CONV3X3: rc2 = dispatch_conv3x3(p, epi, st); break;
        default: rc2 = dispatch_conv1d(p, epi, st); break;
    }
    if (rc2 == PNC_OK && p.ln_out16 && !ln_fused)      // rows span several workgroups (or a generic launch): the LayerNorm kernel
        rc2 = pnc_layernorm(p.out32, p.ldc32, p.M, p.N, p.ln_gamma, p.ln_beta, p.ln_eps, p.ln_out16, p.ldln, nullptr, stream);
    return rc2;
}

PNC_DEFINE_TU_COLLECT(gemm)
