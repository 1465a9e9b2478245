// access_pattern.hip — round 6: is the q-in / o-out stream of the text attention bound by its ACCESS PATTERN?  attn_views_kernel /
// attn_text_kernel read a query row in the MFMA B-fragment layout: lane (row = lane & 31, half = lane >> 5) loads 16 bytes at
// row * ld + head * 128 + ds * 32 + half * 16 — 32 rows x 32 contiguous bytes per wave instruction, four instructions per 128-byte
// head slice — and store O the same way.  Copy kernels, no arithmetic: (A) that pattern, (B) row-contiguous (8 lanes x 16 B per row).
//     hipcc --offload-arch=gfx950 -O2 tools/exp/access_pattern.hip -o tools/exp/access_pattern && tools/exp/access_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void copy(const _Float16* __restrict__ q, _Float16* __restrict__ o, int M, int C, int heads) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x;
    const int head = item % heads, qt = item / heads;
    const long row0 = (long)qt * 128 + wave * 32;
    if (MODE == 0) {
        const long row = row0 + (lane & 31);
        const int half = lane >> 5;
        h8 v[4];
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) v[ds] = *reinterpret_cast<const h8*>(q + row * C + head * 64 + ds * 16 + half * 8);
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) *reinterpret_cast<h8*>(o + row * C + head * 64 + ds * 16 + half * 8) = v[ds];
    } else {
        h8 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = row0 + i * 8 + (lane >> 3);
            v[i] = *reinterpret_cast<const h8*>(q + row * C + head * 64 + (lane & 7) * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = row0 + i * 8 + (lane >> 3);
            *reinterpret_cast<h8*>(o + row * C + head * 64 + (lane & 7) * 8) = v[i];
        }
    }
}

template <int MODE>
static void run(const char* name, int M, int C) {
    _Float16 *q, *o;
    hipMalloc(&q, (size_t)M * C * 2); hipMalloc(&o, (size_t)M * C * 2);
    hipMemset(q, 0, (size_t)M * C * 2);
    const int heads = C / 64, blocks = (M / 128) * heads;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy<MODE>, dim3(blocks), dim3(256), 0, 0, q, o, M, C, heads);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy<MODE>, dim3(blocks), dim3(256), 0, 0, q, o, M, C, heads);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s M=%d C=%d: %7.1f us per launch  %6.0f GB/s\n", name, M, C, ms * 1e3 / reps, (double)M * C * 4 / (ms * 1e-3 / reps) / 1e9);
    hipFree(q); hipFree(o);
}

int main() {
    const int Ms[3] = {196608, 49152, 12288}, Cs[3] = {320, 640, 1280};
    for (int l = 0; l < 3; ++l) {
        run<0>("fragment layout (32 rows x 32 B per instr)", Ms[l], Cs[l]);
        run<1>("row-contiguous (8 rows x 128 B per instr)", Ms[l], Cs[l]);
    }
    return 0;
}
