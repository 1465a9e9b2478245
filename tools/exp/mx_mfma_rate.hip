// mx_mfma_rate.hip — issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 and fp4 operands) against v_mfma_f32_32x32x16_f16 on
// the whole chip: register-resident operands, 4 independent accumulators per wave, 8 waves per CU-sized workgroup.
//     hipcc --offload-arch=gfx950 -O2 tools/exp/mx_mfma_rate.hip -o tools/exp/mx_mfma_rate && tools/exp/mx_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int MODE>      // 0: f16 32x32x16, 1: scaled fp8 32x32x64, 2: scaled fp4 32x32x64
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int t = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + (t & 1); b[i] = 0x38383838; }
    v8h ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(1.0f + (t & 1)); bh[i] = (_Float16)1.0f; }
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c3, 0, 0, 0);
        } else {
            constexpr int F = MODE == 1 ? 0 : 4;
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, F, F, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, F, F, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, F, F, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, F, F, 0, 127, 0, 127);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 512 + t] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma) {
    float* out;
    hipMalloc(&out, 1024 * 512 * 4);
    const int iters = 20000, blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 8 * 4 * iters;
    printf("%-28s %8.2f ms  %8.1f TFLOP/s  %6.2f ns per MFMA and SIMD\n", name, ms, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12,
           ms * 1e6 / (mfmas / 1024.0));
    hipFree(out);
}

int main() {
    run<0>("f16 32x32x16", 2.0 * 32 * 32 * 16);
    run<1>("MX fp8 32x32x64 (scaled)", 2.0 * 32 * 32 * 64);
    run<2>("MX fp4 32x32x64 (scaled)", 2.0 * 32 * 32 * 64);
    return 0;
}
