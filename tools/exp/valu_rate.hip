// valu_rate.hip — round 6: issue cost of the softmax-side VALU instructions of attn_views_kernel on gfx950, per wave-instruction and
// SIMD, with TWO waves per SIMD (the attention kernel's regime: 512-thread blocks, one per CU).  The SQ counters of the level-0
// attention launches say the VALU pipe is the busy one (profiles/round6/attn_pmc_*.txt: VALU active ~ 51 cycles per 32-cycle MFMA);
// this prices the candidates for taking instructions out: packed fp32 fma / add, v_exp_f16, v_dot2_f32_f16 as the row sum.
//     hipcc --offload-arch=gfx950 -O2 tools/exp/valu_rate.hip -o tools/exp/valu_rate && tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// 16 independent chains per lane so that instruction latency never binds; the op under test is inline asm (exact opcode)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
    const int t = threadIdx.x;
    float a[16];
    f2 p[16];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { a[i] = 0.001f * (t + i); p[i] = f2{0.001f * t, 0.002f * i}; u[i] = 0x3c003c00u + t + i; }
    const float k1 = 0.999f, k2 = -0.0001f;
    const f2 kk1 = {0.999f, 0.998f}, kk2 = {-0.0001f, -0.0002f};
    const unsigned ones = 0x3c003c00u;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
            REP16(X)
#undef X
        } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(kk1), "v"(kk2));
            REP16(X)
#undef X
        } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_exp_f16 %0, %0" : "+v"(u[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(k2));
            REP16(X)
#undef X
        } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(kk2));
            REP16(X)
#undef X
        } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[i]), "v"(ones));
            REP16(X)
#undef X
        } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(k1));
            REP16(X)
#undef X
        } else if constexpr (OP == 8) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
            REP16(X)
#undef X
        } else if constexpr (OP == 9) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(kk1));
            REP16(X)
#undef X
        } else if constexpr (OP == 10) {
#define X(i) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(u[i]));
            REP16(X)
#undef X
        } else if constexpr (OP == 11) {
#define X(i) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[i]) : "v"(u[i]), "v"(ones));
            REP16(X)
#undef X
        } else if constexpr (OP == 12) {
#define X(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(u[i]) : "v"(ones));
            REP16(X)
#undef X
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1] + (float)u[i];
    out[blockIdx.x * 512 + t] = s;
}

template <int OP>
static void run(const char* name) {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(512), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x 16 instructions x iters
    const double per_simd = 2.0 * 16 * iters;
    printf("%-34s %8.3f ms  %6.2f ns per wave-instruction and SIMD (= %5.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<4>("v_add_f32");
    run<5>("v_pk_add_f32");
    run<9>("v_pk_mul_f32");
    run<8>("v_max3_f32");
    run<2>("v_exp_f32");
    run<3>("v_exp_f16");
    run<10>("v_exp_f16_sdwa word1");
    run<6>("v_dot2_f32_f16");
    run<11>("v_dot2c_f32_f16");
    run<7>("v_cvt_pk_f16_f32");
    run<12>("v_pk_max_f16");
    return 0;
}
