// gemm_ring3.hip — plain-A GEMM (nn.Linear / 1x1 conv call-sites) on 256 x 320 tiles with THREE A stages.
//
// The two-stage kernels (gemm_kernel.h) give the DMA of a K tile exactly one iteration to land: where one column tile streams A
// from HBM (N = 320: FF2, to_out, the q / qkv projections of level 0) the K loop runs at the DMA's latency, 2.6-3.7 us per tile
// against 1.2 us of MFMAs.  A third whole stage does not fit (3 x 72 KB), but the A operand alone does when W
// rides in gemm_stencil_tile.hip's ring of three HALF tiles (32 of the 64 channels, BN rows x 64 B): 3 x 32 + 3 x 20 = 156 KB.
// A of K tile c + 2 is requested while tile c is consumed (1.5 tiles of flight time under the in-order vmcnt), W half tile q + 3
// while q is — W comes from L2 (every row panel reads it), A is the operand whose latency the third stage hides.
// Pipeline per half tile q (two k-steps of 16, the barrier in the MIDDLE of q's MFMA stream — the stencil kernel's):
//   reads(q, ks1) | MFMA(q, ks0) | wait: W(q+1) [and A of the next K tile] landed, own reads done | s_barrier |
//   DMA: W(q+3) -> stage of q, [A(c+2) -> buffer of c-1 at the first half of K tile c] | reads(q+1, ks0) | MFMA(q, ks1)
// Same tiles, K order, MFMA order per accumulator and epilogue arithmetic as gemm_glds_kernel<PNC_A_PLAIN, 256, 320, 4, 2, 2, false,
// EPI>: bit-identical on every shape tried (tools/exp/ring3_ab.py).
//
// NOT SHIPPED (round 6): measured 0-15 % SLOWER than the two-stage kernels at every plain-A shape of config 3
// (profiles/round6/ring3_ab_r6.log: L0 FF2 258 -> 267 us, L0 q 72 -> 78, L1 FF2 209 -> 232, L2 QKV 131 -> 152; level-2 C x C shapes
// equal) — so the K loop of the N = 320 class is NOT waiting on A's latency: 1.5 K tiles of flight time change nothing, the half-tile
// ring's second barrier per K tile costs what it costs in the stencil kernel.  To try it again: copy this file to panacea_amd/csrc/,
// add it to build.SOURCES, call pnc_tu_collect_gemm_ring3 from misc.hip, and in gemm_plain.hip's dispatch_plain() try
// `plain_ring3_ok(p, epi)` / `dispatch_plain_ring3(p, epi, st)` before the persistent kernel (PNC_OPT_GEMM_PERSIST bit 2).
#include "gemm_kernel.h"

namespace pnc_gemm {

template <int N>
__device__ __forceinline__ void vm_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// (wave-uniform) counted wait: at most n of this wave's DMA instructions outstanding
__device__ __forceinline__ void vm_wait(int n) {
    switch (n) {
        case 0: vm_wait_n<0>(); break;
        case 1: vm_wait_n<1>(); break;
        case 2: vm_wait_n<2>(); break;
        case 3: vm_wait_n<3>(); break;
        case 4: vm_wait_n<4>(); break;
        case 5: vm_wait_n<5>(); break;
        case 6: vm_wait_n<6>(); break;
        default: vm_wait_n<7>(); break;
    }
}

template <int NI, unsigned EPI>
__global__ __launch_bounds__(512) void ring3_kernel(const PncGemmParams pin, const int group_m) {
    const PncGemmParams& p = pin;
    constexpr int BM = 256, BN = NI * 64, NW = 8, WGN = 2, MI = 2;
    constexpr int ABYTES = BM * 128, A_IT = BM / (NW * 8);  // a K tile of A: 32 pieces of 1 KB (8 rows x 128 B), 4 per wave
    constexpr int WHB = BN * 64;                            // bytes of a W half tile: BN rows x 32 channels
    constexpr int WBLK = WHB / 1024;                        // its 1-KB DMA pieces (16 W rows x 64 B each): 20 / 16
    constexpr int W_IT = (WBLK + NW - 1) / NW;
    constexpr int ENI = 2, EPITCH = ENI * 32 + 4;
    static_assert(3 * ABYTES + 3 * WHB <= 160 * 1024, "LDS budget");
    static_assert(3 * ABYTES + 3 * WHB >= NW * (32 * EPITCH * 4 + 64 * 8), "epilogue staging (+ LayerNorm row sums) fits the operand rings");
    static_assert(W_IT + A_IT <= 7, "vm_wait's range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const aring = smem;
    char* const wring = smem + 3 * ABYTES;

    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);
    const int tiles_n = p.N / BN, tiles_m = p.M / BM, ntile = tiles_m * tiles_n;
    const int tile = xcd_remap(blockIdx.x, ntile);
    int tn, tm;
    if (group_m > 0) {                                      // gemm_glds_kernel's grouped order
        const int width = group_m * tiles_n;
        const int gid = tile / width, first_m = gid * group_m;
        const int gsz = min(tiles_m - first_m, group_m);
        const int in = tile - gid * width;
        tm = first_m + in % gsz; tn = in / gsz;
    } else {
        tn = tile % tiles_n; tm = tile / tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const buffer_rsrc_t rs_a = make_rsrc(A + (int64_t)m0 * p.lda, 0x7FFFFF00u);
    const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);
    // ---- A DMA: piece (wave, i) = tile rows i*64 + wave*8 .. +7; lane l fills slot (l&7) of its row with the source chunk
    // slot ^ ((row>>1)&7) (common.h lds_off128); the K tile enters as the scalar offset
    unsigned aoff[A_IT];
    {
        const int srow = wave * 8 + (lane >> 3), schunk = (lane & 7) ^ ((srow >> 1) & 7);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) aoff[i] = (unsigned)((i * (NW * 8) + srow) * p.lda + schunk * 8) * 2u;
    }
    auto issue_a = [&](int c, int buf) {
        char* sa = aring + buf * ABYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16_buf(rs_a, aoff[i], (unsigned)c << 7, sa + i * (NW * 1024));
    };
    // ---- W DMA: half tile k = k offset 32 k.  LDS row R (128 B) = W rows 2R, 2R+1; slot = (n&1)*4 + (c ^ ((R>>1)&3)),
    // c = 16-byte chunk of the 64-byte half row (gemm_stencil_tile.hip's layout)
    const int nW = WBLK / NW + (wave < (WBLK % NW) ? 1 : 0);      // DMA instructions of this wave per half tile
    unsigned woff[W_IT];
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
        const int R = (wave + NW * i) * 8 + (lane >> 3), slot = lane & 7;
        const int nl = 2 * R + (slot >> 2);
        const int c4 = (slot & 3) ^ ((R >> 1) & 3);
        woff[i] = (unsigned)(nl * p.ldw + c4 * 8) * 2u;
    }
    auto issue_w = [&](int k, int stage) {
        char* sb = wring + stage * WHB + wave * 1024;
#pragma unroll
        for (int i = 0; i < W_IT; ++i)
            if (wave + NW * i < WBLK) glds16_buf(rs_w, woff[i], (unsigned)k << 6, sb + i * (NW * 1024));
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fk = lane >> 5;
    const int a_row = (wm * 64 + frow) * 128, a_swz = (frow >> 1) & 7;          // (row>>1)&7: blocks of 32 rows shift it by 16
    const int b_row = ((wn * (NI * 32) + frow) >> 1) * 128 + ((frow & 1) << 6);
    const int b_swz = (frow >> 2) & 3;
    // fragments of one k-step (16 channels) of (A buffer, half, W stage) -> register buffer b; explicit ds_read_b128 with the
    // waits written out below (gemm_stencil_tile.hip has the reason)
    half8v af[2][MI], bf[2][NI];
    auto frags = [&](int abuf, int half, int stage, int ks, auto b_) {
        constexpr int b = decltype(b_)::value;
        const int c4 = ks * 2 + fk;
        const unsigned a_addr = (unsigned)(uintptr_t)(aring + abuf * ABYTES + a_row + (((half * 4 + c4) ^ a_swz) << 4));
        asm volatile("ds_read_b128 %0, %1" : "=v"(af[b][0]) : "v"(a_addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[b][1]) : "v"(a_addr));
        const unsigned b_addr = (unsigned)(uintptr_t)(wring + stage * WHB + b_row + ((c4 ^ b_swz) << 4));
#define PNC_RING3_RD_B(J)                                                                                              \
    if constexpr (NI > J)                                                                                              \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[b][J < NI ? J : 0]) : "v"(b_addr), "n"(J * 16 * 128));
        PNC_RING3_RD_B(0) PNC_RING3_RD_B(1) PNC_RING3_RD_B(2) PNC_RING3_RD_B(3) PNC_RING3_RD_B(4)
#undef PNC_RING3_RD_B
    };
    auto mfmas = [&](auto b_) {
        constexpr int b = decltype(b_)::value;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b][i], bf[b][j], acc[i][j], 0, 0, 0);
    };
    const std::integral_constant<int, 0> B0{};
    const std::integral_constant<int, 1> B1{};

    // DMA order per iteration: [W half tile, A tile]; the in-order counter then lets the A tile issued one or two iterations ago
    // stay in flight while the older W half tile is waited for
    const int nslices = p.K >> 6, nq = nslices * 2;
    issue_a(0, 0);
    issue_w(0, 0);
    if (nq > 1) { issue_w(1, 1); vm_wait(nW); }               // (nq >= 2 always: K >= 64)
    __builtin_amdgcn_s_barrier();
    if (nq > 2) issue_w(2, 2);
    if (nslices > 1) issue_a(1, 1);
    frags(0, 0, 0, 0, B0);
    int st = 0, gs = 0, r = 0, ab = 0;                        // W stage of half tile q, K tile, half, A buffer of the K tile
    for (int q = 0; q < nq; ++q) {
        frags(ab, r, st, 1, B1);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MI + NI) : "memory");       // buffer 0 (the older reads) is in
        __builtin_amdgcn_sched_barrier(0);
        mfmas(B0);
        __builtin_amdgcn_sched_barrier(0);
        const int r1 = r ^ 1, gs1 = gs + r;
        const int ab1 = r ? (ab == 2 ? 0 : ab + 1) : ab;
        const int st1 = (st == 2) ? 0 : st + 1;
        // in flight after the wait: W(q+2) and the A tile issued behind W(q+1) or W(q+2) — K tile gs+1 (first half) / gs+2 (second);
        // W(q+1) has landed, and at the second half so has A of K tile gs+1, read from the next iteration on
        if (q + 2 < nq) vm_wait(nW + (((r ? gs + 2 : gs + 1) < nslices) ? A_IT : 0));
        else vm_wait_n<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave has read everything it needs of stage st
        __builtin_amdgcn_s_barrier();
        // W stage st was read by half tile q, A buffer (ab + 2) % 3 by K tile gs - 1: all waves are past both
        if (q + 3 < nq) issue_w(q + 3, st);
        if (r == 0 && gs + 2 < nslices) issue_a(gs + 2, ab == 0 ? 2 : ab - 1);
        if (q + 1 < nq) frags(ab1, r1, st1, 0, B0);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(B1);
        __builtin_amdgcn_sched_barrier(0);
        st = st1; r = r1; gs = gs1; ab = ab1;
    }
    __syncthreads();                            // every wave is done with the operand rings

    // ------------------------------ epilogue (gemm_glds_kernel's) ------------------------------
    const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
    if constexpr ((EPI & E_VT) != 0) {
        if (n0 >= p.n_split) { epi_vt<MI, NI>(p, acc, lane, mw, nw); return; }
    }
    float* ep = reinterpret_cast<float*>(smem) + wave * (32 * EPITCH);
    if constexpr ((EPI & E_LN) != 0) {
        float2* lnb = reinterpret_cast<float2*>(reinterpret_cast<float*>(smem) + NW * (32 * EPITCH));
        epi_fast<MI, NI, EPI>(p, acc, ep, lane, RowLinear{mw}, nw, p.N, lnb + wave * 64, lnb + (wave ^ 1) * 64);
    } else {
        epi_fast<MI, NI, (EPI & ~E_VT)>(p, acc, ep, lane, RowLinear{mw}, nw, (EPI & E_VT) ? p.n_split : p.N);
    }
}

// the three-A-stage kernel serves this problem (PNC_OPT_GEMM_PERSIST bit 2)
bool plain_ring3_ok(const PncGemmParams& p, unsigned epi) {
    if ((pnc_get_option(PNC_OPT_GEMM_PERSIST) & 4) == 0 || p.a_mode != PNC_A_PLAIN) return false;
    if ((p.M % 256) || (p.N % 320) || (p.K % 64) || p.K < 128 || p.A_lo) return false;
    if ((epi & E_LN) && p.N != 320) return false;
    if ((epi & E_VT) && ((p.n_split % 320) || (p.M % 8) || (p.t_rows % 8))) return false;
    return true;
}

template <unsigned EPI>
static int launch_ring3(const PncGemmParams& p, hipStream_t st) {
    constexpr int NI = 5, lds = 3 * 256 * 128 + 3 * NI * 64 * 64;
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto kern = ring3_kernel<NI, EPI>;
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    const int tiles_n = p.N / 320, tiles_m = p.M / 256;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? gopt : (tiles_n > 8 ? 4 : 0);
    if (group_m > tiles_m) group_m = tiles_m;
    if (tiles_n < 2) group_m = 0;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, p, group_m);
    return pnc_launch_status();
}

// PNC_EINVAL: no instantiation for this epilogue variant (the caller falls through to the two-stage kernels)
int dispatch_plain_ring3(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    switch (epi) {
        case E_O16: return launch_ring3<E_O16>(p, st);
        case E_R1 | E_O16: return launch_ring3<E_R1 | E_O16>(p, st);
        case E_R1 | E_O32 | E_LN: return launch_ring3<E_R1 | E_O32 | E_LN>(p, st);
        default: return PNC_EINVAL;
    }
}

}  // namespace pnc_gemm

PNC_DEFINE_TU_COLLECT(gemm_ring3)
