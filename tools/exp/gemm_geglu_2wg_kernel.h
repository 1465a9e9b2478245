// EXPERIMENT RECORD (round 4), not compiled into libpanacea_hip.so: the GEGLU (FF1) GEMM as two independent persistent 4-wave
// workgroups per CU.  To reproduce: paste into panacea_amd/csrc/gemm_kernel.h in front of launch_geglu_persist, dispatch from
// gemm_plain.hip's E_GEGLU case with `if (tc.tile == T_256x256 && geglu_2wg_ok(p)) return launch_geglu_2wg(p, st);`, run
// tools/exp/ff1_2wg_ab.py (PNC_OPT_GEMM_PERSIST = 5).  Bit-identical to the shipped kernels (tests passed); measured slower —
// profiles/round4/ff1_two_workgroups_ab_r4i.txt.
// GEGLU GEMM as TWO independent persistent workgroups per CU (round 4 experiment -> PNC_OPT_GEMM_PERSIST bit 2): 4 waves (2 x 2),
// tile 128 x 256, the same 64 x 128 wave tile and register epilogue as gemm_geglu_persist_kernel.  Why: at K = 320 the GEGLU
// epilogue (gate arithmetic on the VALU + table reads) is as long as the main loop, and the eight waves of the one-workgroup
// kernel run the two in lockstep — its barrier per K tile re-aligns the two waves of every SIMD, exactly the pattern that cost the
// view attention 8 % until its workgroup was split in two.  Two workgroups share nothing but the
// CU: one's epilogue runs under the other's MFMAs.  The price is the one round 3 measured for residual GEMMs: W is staged once
// per 128 rows instead of once per 256 (1.5x the LDS-DMA pieces per MFMA).
// LDS per workgroup: ring of two K HALF tiles (32 channels: A 128 x 64 B + W 256 x 64 B = 24 KB each) + Phi table 16 KB + staging
// 4 x 2 KB = 72 KB.  Half-tile rows are 64 B: LDS row R (128 B) holds operand rows 2R and 2R + 1, chunk c of a row in slot
// (row & 1) * 4 + (c ^ ((R >> 1) & 3)) — gemm_stencil_tile.hip's W layout, conflict-free for the 16-lane groups of ds_read_b128.
// Same k order, same MFMA order per accumulator, same epilogue arithmetic: bit-identical to the other two FF1 kernels.
__device__ __forceinline__ int lds_off64(int row, int chunk) {      // byte offset of 16-byte chunk `chunk` (0..3) of operand row `row`
    const int R = row >> 1;
    return R * 128 + ((((row & 1) << 2) + (chunk ^ ((R >> 1) & 3))) << 4);
}
template <int TAG>      // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(256, 2) void gemm_geglu_2wg_kernel(const PncGemmParams pin, const float* __restrict__ phi_g, const int group_m) {
    PncGemmParams p = pin;
    constexpr int BM = 128, BN = 256, WGM = 2, WGN = 2, NW = 4, MI = 2, NI = 4, HK = 32;
    constexpr int A_PC = BM * 64 / 1024, B_PC = BN * 64 / 1024;          // 1-KB DMA pieces per half tile: 8 + 16
    constexpr int A_IT = A_PC / NW, B_IT = B_PC / NW;                     // per wave: 2 + 4
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;              // 8 KB + 16 KB
    extern __shared__ __attribute__((aligned(16))) char smem[];           // the ring: 2 x 24 KB, the only memory LDS-DMA writes
    __shared__ __attribute__((aligned(16))) float s_phi[PHI_BYTES / 4];
    __shared__ __attribute__((aligned(16))) half_t s_stage[NW][32 * 32];
    const half_t* __restrict__ A = reinterpret_cast<const half_t*>(p.A);
    const half_t* __restrict__ Wt = reinterpret_cast<const half_t*>(p.W);
    const int tiles_n = p.N / BN, tiles_m = p.M / BM, ntile = tiles_m * tiles_n, nk = p.K / HK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 31, fk = lane >> 5;
    auto tile_origin = [&](int v, int& m0, int& n0) {
        const int tile = xcd_remap(v, ntile);
        int tn, tm;
        if (group_m > 0) {
            const int width = group_m * tiles_n;
            const int gid = tile / width, first_m = gid * group_m;
            const int gsz = min(tiles_m - first_m, group_m);
            const int in = tile - gid * width;
            tm = first_m + in % gsz; tn = in / gsz;
        } else {
            tn = tile % tiles_n; tm = tile / tiles_n;
        }
        m0 = tm * BM; n0 = tn * BN;
    };
    // DMA piece pi = wave + NW * i covers LDS rows 8 pi .. 8 pi + 7; lane l fills slot (l & 7) of LDS row 8 pi + (l >> 3): operand row
    // 2 R + (slot >> 2), source chunk (slot & 3) ^ ((R >> 1) & 3)
    unsigned aoff[A_IT], woff[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int R = (wave + NW * i) * 8 + (lane >> 3), slot = lane & 7;
        aoff[i] = (unsigned)((2 * R + (slot >> 2)) * p.lda + ((slot & 3) ^ ((R >> 1) & 3)) * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int R = (wave + NW * i) * 8 + (lane >> 3), slot = lane & 7;
        woff[i] = (unsigned)((2 * R + (slot >> 2)) * p.ldw + ((slot & 3) ^ ((R >> 1) & 3)) * 8) * 2u;
    }
    auto issue = [&](int m0, int n0, int ht, int stage) {
        const buffer_rsrc_t rs_a = make_rsrc(A + (int64_t)m0 * p.lda, 0x7FFFFF00u);
        const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * p.ldw, 0x7FFFFF00u);
        char* sa = smem + stage * STAGE;
        char* sb = sa + A_BYTES;
        const unsigned ks = (unsigned)ht * (HK * 2);
#pragma unroll
        for (int i = 0; i < A_IT; ++i) glds16_buf(rs_a, aoff[i], ks, sa + (wave + NW * i) * 1024);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) glds16_buf(rs_w, woff[i], ks, sb + (wave + NW * i) * 1024);
    };
    for (int i = tid; i < PHI_BYTES / 16; i += 64 * NW)
        reinterpret_cast<f32x4*>(s_phi)[i] = reinterpret_cast<const f32x4*>(phi_g)[i];
    f32x16 acc[MI][NI];
    auto compute = [&](int stage) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        half8v af[2][MI], bf[2][NI];
        auto frags = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[b][i] = *reinterpret_cast<const half8v*>(sa + lds_off64(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bf[b][j] = *reinterpret_cast<const half8v*>(sb + lds_off64(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
        };
        frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < HK / 16; ++ks) {
            if (ks + 1 < HK / 16) frags(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int v = blockIdx.x;
    if (v >= ntile) return;
    int m0, n0, sp = 0;
    tile_origin(v, m0, n0);
    float pb[NI], pbn[NI];
    auto load_bias = [&](int n0_, float (&dst)[NI]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) dst[j] = p.bias ? p.bias[n0_ + wn * (NI * 32) + j * 32 + (lane & 31)] : 0.0f;
    };
    load_bias(n0, pb);
    issue(m0, n0, 0, 0);
    while (true) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(m0, n0, kt + 1, (sp + kt + 1) & 1);
            compute((sp + kt) & 1);
            __syncthreads();
        }
        const int ls = (sp + nk - 1) & 1;
        const int vn = v + gridDim.x;
        int m1 = 0, n1 = 0;
        if (vn < ntile) {
            tile_origin(vn, m1, n1);
            load_bias(n1, pbn);
            issue(m1, n1, 0, ls ^ 1);
        }
        {   // epi_geglu's register path, slab by slab (gemm_geglu_persist_kernel's: bit-identical)
            half_t* out16 = reinterpret_cast<half_t*>(p.out16);
            typedef half_t __attribute__((may_alias)) half_st;
            typedef int4 __attribute__((may_alias)) int4_st;
            half_st* sb = reinterpret_cast<half_st*>(&s_stage[wave][0]);
            const int c = lane & 31, cl = lane & 3, rl = lane >> 2;
            const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
            static_for<NI / 2>([&](auto jc_) {
                constexpr int jc = decltype(jc_)::value * 2;
                const float bv = pb[jc], bg = pb[jc + 1];
                const int ncol0 = (nw + jc * 32) >> 1;
                static_for<MI>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    float gx[16], fr[16];
                    int ix[16];
                    float2 e[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        gx[r] = acc[i][jc + 1][r] + bg;
                        float t = fmaf(gx[r], PHI_SCALE, -PHI_X0 * PHI_SCALE);
                        t = __builtin_amdgcn_fmed3f(t, 0.0f, (float)PHI_N - 0.001f);
                        ix[r] = (int)t;
                        fr[r] = t - (float)ix[r];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) e[r] = *reinterpret_cast<const float2*>(s_phi + 2 * ix[r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float prod = (acc[i][jc][r] + bv) * (gx[r] * fmaf(fr[r], e[r].y, e[r].x));
                        asm("" : "+v"(prod));
                        sb[mfma32_row(r, lane) * 32 + c] = (half_t)prod;
                    }
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        const int row = ps * 16 + rl;
                        const int4 v4 = *reinterpret_cast<const int4_st*>(sb + row * 32 + cl * 8);
                        *reinterpret_cast<int4_st*>(out16 + (int64_t)(mw + i * 32 + row) * p.ldc16 + ncol0 + cl * 8) = v4;
                    }
                });
            });
        }
        if (vn >= ntile) break;
        v = vn; m0 = m1; n0 = n1; sp = ls ^ 1;
#pragma unroll
        for (int j = 0; j < NI; ++j) pb[j] = pbn[j];
    }
}

static inline int launch_geglu_2wg(const PncGemmParams& p, hipStream_t st) {
    constexpr int lds = 2 * (128 + 256) * 64;
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_geglu_2wg_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    int rc = PNC_OK;
    const float* phi = phi_table_device(st, &rc);
    if (rc != PNC_OK) return rc;
    const int tiles_n = p.N / 256, tiles_m = p.M / 128, tiles = tiles_m * tiles_n;
    const int gopt = pnc_get_option(PNC_OPT_GEMM_GROUP_M);
    int group_m = gopt > 0 ? 2 * gopt : (tiles_n > 8 ? 8 : 0);        // 8 panels of 128 rows = the 4 panels of 256 of the other kernels
    if (group_m > tiles_m) group_m = tiles_m;
    if (tiles_n < 2) group_m = 0;
    static std::atomic<int> ncu_of[64];
    int ncu = ncu_of[dev & 63].load(std::memory_order_relaxed);
    if (ncu == 0) {
        int v = 0;
        ncu = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        ncu_of[dev & 63].store(ncu, std::memory_order_relaxed);
    }
    const int blocks = tiles < 2 * ncu ? tiles : 2 * ncu;
    hipLaunchKernelGGL(gemm_geglu_2wg_kernel<0>, dim3(blocks), dim3(256), lds, st, p, phi, group_m);
    return pnc_launch_status();
}
static inline bool geglu_2wg_ok(const PncGemmParams& p) {
    return (pnc_get_option(PNC_OPT_GEMM_PERSIST) & 4) != 0 && p.geglu && p.a_mode == PNC_A_PLAIN && !p.A_lo && !p.out16_lo && p.out16 &&
           (p.M % 128) == 0 && (p.N % 256) == 0 && (p.K % 32) == 0 && p.K >= 64 && (p.M / 128) * (p.N / 256) >= 1024;
}

