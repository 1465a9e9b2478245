// ffchain_probe.hip — where does pnc_ff_chain_f16 spend its time?  Standalone (no torch):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I panacea_amd/csrc tools/exp/ffchain_probe.hip panacea_amd/lib/gemm.o \
//           panacea_amd/lib/misc.o ... -o tools/exp/ffchain_probe && tools/exp/ffchain_probe
// Times ff_chain_kernel<ABL> for the ablation masks of the kernel (1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no GEGLU,
// 16 no barriers, 32 no prologue loads / epilogue stores) at the level-0 shape of BASELINE config 3 (M = 196 608, C = 320).
#include "ff_chain_kernel.hip"
#include <cstdio>
#include <vector>

template <int ABL>
static float run(const PncFfChainParams& p, const float*, int iters) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ff_chain_kernel<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(ff_chain_kernel<ABL>, dim3(p.M / ROWS), dim3(256), LDS_BYTES, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(ff_chain_kernel<ABL>, dim3(p.M / ROWS), dim3(256), LDS_BYTES, 0, p);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms * 1e3f / iters;
}

int main() {
    const int M = 196608, C = 320, inner = 1280;
    PncFfChainParams p{};
    float *x, *o32, *gam, *bet, *b1, *b2, *phi;
    void *tape, *o16, *lo;
    hipMalloc(&x, (size_t)M * C * 4); hipMalloc(&o32, (size_t)M * C * 4); hipMalloc(&o16, (size_t)M * C * 2); hipMalloc(&lo, (size_t)M * C);
    hipMalloc(&gam, C * 4); hipMalloc(&bet, C * 4); hipMalloc(&b1, 2 * inner * 4); hipMalloc(&b2, C * 4); hipMalloc(&phi, 16384);
    const size_t tb = (size_t)3 * (inner / 32) * 20480;
    hipMalloc(&tape, tb);
    hipMemset(x, 0, (size_t)M * C * 4); hipMemset(tape, 0, tb); hipMemset(gam, 0, C * 4); hipMemset(bet, 0, C * 4);
    hipMemset(b1, 0, 2 * inner * 4); hipMemset(b2, 0, C * 4); hipMemset(phi, 0, 16384);
    {   // non-trivial data: random-ish bits in x and the tape (DVFS: zero data clocks higher)
        std::vector<unsigned short> h(tb / 2);
        unsigned s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x2C00 | ((s >> 16) & 0x83FF)); }
        hipMemcpy(tape, h.data(), tb, hipMemcpyHostToDevice);
        std::vector<float> hx((size_t)M * C);
        for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
        hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> g1(C, 1.0f);
        hipMemcpy(gam, g1.data(), C * 4, hipMemcpyHostToDevice);
    }
    p.x32 = x; p.ldx = C; p.M = M; p.C = C; p.inner = inner; p.ln_eps = 1e-5f; p.ln_gamma = gam; p.ln_beta = bet; p.tape = tape;
    p.b1 = b1; p.b2 = b2; p.out32 = nullptr; p.ldo32 = C; p.ldo16 = C; p.out16 = o16; p.out16_lo = lo; p.out_lo_fmt = PNC_LO_E4M3;
    p.struct_bytes = sizeof(p);
    const int it = 10;
    printf("full kernel                                  %8.1f us\n", run<0>(p, phi, it));
    printf("no tape DMA in the loop                (1)   %8.1f us\n", run<1>(p, phi, it));
    printf("no MFMAs                               (2)   %8.1f us\n", run<2>(p, phi, it));
    printf("no GEGLU arithmetic                    (8)   %8.1f us\n", run<8>(p, phi, it));
    printf("no barriers, no DMA                    (17)  %8.1f us\n", run<17>(p, phi, it));
    printf("no prologue loads / epilogue stores    (32)  %8.1f us\n", run<32>(p, phi, it));
    printf("no DMA, no MFMA                        (3)   %8.1f us\n", run<3>(p, phi, it));
    printf("no DMA, no fragment reads, no barriers (21)  %8.1f us  (MFMA + GEGLU + pro/epilogue)\n", run<21>(p, phi, it));
    printf("MFMA only                              (61)  %8.1f us  (no DMA, reads, GEGLU, barriers, pro/epilogue)\n", run<61>(p, phi, it));
    printf("DMA + barriers only                    (46)  %8.1f us  (no MFMA, reads, GEGLU, pro/epilogue)\n", run<46>(p, phi, it));
    p.out32 = o32; p.out16 = nullptr; p.out16_lo = nullptr;
    printf("full kernel, fp32 output                     %8.1f us\n", run<0>(p, phi, it));
    return 0;
}
