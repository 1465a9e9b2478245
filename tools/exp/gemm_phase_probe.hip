// gemm_phase_probe.hip — round 5, VERDICT r4 item 1: the plain-A GEMM main loop in three schedules, standalone (no torch).
//
//   C[M,N] (fp16) = A[M,K] x W[N,K]^T, fp16 operands, fp32 accumulate, 256 x BN tiles (BN = 256 | 320), BK = 64, 8 waves as 4 x 2,
//   v_mfma_f32_32x32x16_f16, operands HBM -> LDS by buffer_load ... lds into XOR-swizzled 128-byte rows — the geometry, DMA
//   pieces and fragment reads of gemm_kernel.h.  What differs is the SCHEDULE of the K loop:
//
//   VAR 0  "B"  the shipped loop: the next K tile's DMA issued in one block (waves 4-7 in the middle of their MFMA stream),
//               4 k-steps of {fragment reads, MI x NI MFMAs}, one __syncthreads (vmcnt(0) + barrier) per K tile.
//   VAR 1  "P"  phased: a K tile is four phases {fragment reads of one k-step + a share of the next tile's DMA | s_barrier |
//               MI x NI MFMAs at raised priority | s_barrier}; waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one
//               wave is in its MFMA cluster while the other reads / issues DMA (the two-group schedule of the HIP guide's
//               "256^2 8-phase template", on this kernel's 4 x 2 wave grid and full-tile stages); vmcnt(0) once per K tile, in
//               phase 3, a whole phase after the last DMA issue.
//               (first run, profiles/round5/gemm_phase_probe_r5a.log: priority flips around the MFMA clusters are flat — dropped)
//   VAR 2  "P01" P with the next tile's DMA in phases 0-1 only; VAR 3 "P0": all of it in phase 0 (more time to land before phase 3's wait)
//   VAR 6  "Pm"  P with the DMA pieces placed INSIDE the MFMA clusters (one after every third MFMA), the L parts carry reads only
//   VAR 7  "Pn"  P without the group offset (every wave in the same phase: isolates the stagger)
//   VAR 4  "U"  P on a ring of four K-HALF units (32 k = 64-byte LDS rows, 16 rows per DMA piece) instead of two full K tiles: same
//               LDS bytes, but a unit is recycled two phases after its last read, so the DMA is spread evenly over ALL phases
//               (2-3 pieces each), stays 2-3 units ahead and is waited for with a COUNTED vmcnt (never 0 in the steady state).
//   Timing: "warm" = one operand set re-used by every launch (A comes from the 256 MB Infinity Cache after the first launch);
//   "cold" = the launches rotate over operand sets that together exceed the Infinity Cache, so A streams from HBM (the network's
//   case for the big activations).
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I panacea_amd/csrc tools/exp/gemm_phase_probe.hip -o tools/exp/gemm_phase_probe
//   run:    tools/exp/gemm_phase_probe            (on the GPU box; verifies every variant against a host reference first)
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

int pnc_get_option(int) { return 0; }

constexpr int BK = 64;

template <int BN, int VAR>
__global__ __launch_bounds__(512) void gemm_probe(const half_t* __restrict__ A, const half_t* __restrict__ Wt, half_t* __restrict__ C,
                                                  int M, int N, int K, int store) {
    constexpr int BM = 256, WGM = 4, WGN = 2, NW = 8;
    constexpr int MI = BM / WGM / 32, NI = BN / WGN / 32;
    constexpr int RPI = NW * 8, A_IT = BM / RPI, B_IT = BN / RPI, LOADS = A_IT + B_IT;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = N / BN, tiles_m = M / BM, ntile = tiles_m * tiles_n;
    const int tile = xcd_remap(blockIdx.x, ntile);
    int tn, tm;
    const int group_m = tiles_n > 8 ? min(4, tiles_m) : 0;
    if (group_m > 0) {
        const int width = group_m * tiles_n;
        const int gid = tile / width, first_m = gid * group_m;
        const int gsz = min(tiles_m - first_m, group_m);
        const int in = tile - gid * width;
        tm = first_m + in % gsz; tn = in / gsz;
    } else {
        tn = tile % tiles_n; tm = tile / tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN, nt = K / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int srow = wave * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((srow >> 1) & 7);
    const buffer_rsrc_t rs_a = make_rsrc(A + (int64_t)m0 * K, 0x7FFFFF00u);
    const buffer_rsrc_t rs_w = make_rsrc(Wt + (int64_t)n0 * K, 0x7FFFFF00u);
    unsigned aoff[A_IT], woff[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) aoff[i] = (unsigned)((i * RPI + srow) * K + schunk * 8) * 2u;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) woff[i] = (unsigned)((i * RPI + srow) * K + schunk * 8) * 2u;
    // DMA piece q (0 .. LOADS-1) of K tile kt into `stage`
    auto issue_piece = [&](int q, int kt, int stage) {
        char* sa = smem + stage * STAGE + wave * 1024;
        char* sb = sa + A_BYTES;
        const unsigned ks = (unsigned)kt * (BK * 2);
        if (q < A_IT) glds16_buf(rs_a, aoff[q], ks, sa + q * (RPI * 128));
        else glds16_buf(rs_w, woff[q - A_IT], ks, sb + (q - A_IT) * (RPI * 128));
    };
    auto issue_tile = [&](int kt, int stage) {
#pragma unroll
        for (int q = 0; q < LOADS; ++q) issue_piece(q, kt, stage);
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31, fk = lane >> 5;
    half8v af[MI], bf[NI];
    auto frags = [&](int stage, int ks) {
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            af[i] = *reinterpret_cast<const half8v*>(sa + lds_off128(wm * (MI * 32) + i * 32 + frow, ks * 2 + fk));
#pragma unroll
        for (int j = 0; j < NI; ++j)
            bf[j] = *reinterpret_cast<const half8v*>(sb + lds_off128(wn * (NI * 32) + j * 32 + frow, ks * 2 + fk));
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    if constexpr (VAR == 4 || VAR == 5) {
        // ---- ring of four k-half units.  Unit rows R = 0 .. 255 (A), 256 .. 255 + BN (W); row R at byte 64 R of the unit, its 16-byte
        // chunk c (8 k) in slot c ^ ((R >> 2) & 3): the 16 lanes of a ds_read_b128 group (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
        // of a 32-row block, one chunk) then cover all 16 slots of the four 64-byte rows that share a 256-byte bank row.
        constexpr bool STAGGER = VAR != 5;
        constexpr int UROWS = BM + BN, UNIT = UROWS * 64;
        constexpr int FULLP = UROWS / 128;                 // full 16-row pieces per wave and unit (4)
        constexpr bool HALFP = (UROWS % 128) != 0;         // + one 8-row piece of lanes 0-31 (BN = 320: 576 rows = 8 x (4 x 16 + 8))
        static_assert(UROWS % 128 == 0 || UROWS % 128 == 64, "rows per wave: whole pieces, or whole pieces + half a piece");
        constexpr int PU = FULLP + (HALFP ? 1 : 0);        // DMA instructions per wave and unit
        constexpr int S0 = (PU + 1) / 2, S1 = PU - S0;     // issued in the odd / even phase
        static_assert(4 * UNIT <= 2 * STAGE, "the ring takes the two stages' bytes");
        const int nu = K / 32;
        const int prow = lane >> 2, pch = lane & 3;
        unsigned poff[PU];                                 // per-lane byte offset of piece q from its operand's tile origin (k = 0)
        bool pisw[PU];
#pragma unroll
        for (int q = 0; q < PU; ++q) {
            const int R = q < FULLP ? (q * 8 + wave) * 16 + prow : FULLP * 128 + wave * 8 + prow;     // unit row of this lane
            const int c = pch ^ ((R >> 2) & 3);                                                       // the chunk that belongs in slot pch
            pisw[q] = R >= BM;
            poff[q] = (unsigned)((pisw[q] ? R - BM : R) * K + c * 8) * 2u;
        }
        auto issue_u = [&](int q, int u) {                 // piece q of unit u
            char* dst = smem + (u & 3) * UNIT + (q < FULLP ? (q * 8 + wave) * 1024 : FULLP * 8192 + wave * 512);
            const unsigned ks = (unsigned)u * 64u;         // 32 k = 64 bytes along the row
            // (A / W is a property of the PIECE for whole pieces: rows of one piece never straddle row 256)
            if (q < FULLP) {
                if (q < BM / 128) glds16_buf(rs_a, poff[q], ks, dst);
                else glds16_buf(rs_w, poff[q], ks, dst);
            } else if (lane < 32) {
                glds16_buf(rs_w, poff[q], ks, dst);
            }
        };
        auto frags_u = [&](int u, int ks) {
            const char* ub = smem + (u & 3) * UNIT;
            const int c = ks * 2 + fk;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int R = wm * (MI * 32) + i * 32 + frow;
                af[i] = *reinterpret_cast<const half8v*>(ub + R * 64 + ((c ^ ((R >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int R = BM + wn * (NI * 32) + j * 32 + frow;
                bf[j] = *reinterpret_cast<const half8v*>(ub + R * 64 + ((c ^ ((R >> 2) & 3)) << 4));
            }
        };
        const int grp = wave >> 2;
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (u < nu) {
#pragma unroll
                for (int q = 0; q < PU; ++q) issue_u(q, u);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();
        // phase phi = 2 u + ks.  DMA: phi = 2 v + 1 issues share 0 of unit v + 3, phi = 2 v + 2 share 1 — into the slot of unit v - 1,
        // whose last reads (phase 2 v - 1) every wave of both groups has waited for two barriers earlier.  Wait: unit u is first read
        // in phase 2 u; the counted wait sits in phase 2 u - 1 BEFORE its first barrier (the other group is one barrier away from
        // reading), after that phase's DMA issue: units u + 1 (whole) and u + 2 (share 0) may stay in flight.
        for (int u = 0; u < nu; ++u) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frags_u(u, ks);
                if (ks == 1) {
                    if (u + 3 < nu) {
#pragma unroll
                        for (int q = 0; q < S0; ++q) issue_u(q, u + 3);
                    }
                    // unit u + 1 must have landed before this phase's first barrier
                    if (u + 3 < nu) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PU + S0) : "memory");
                    else if (u + 2 < nu) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PU) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else if (u >= 1 && u + 2 < nu) {
#pragma unroll
                    for (int q = S0; q < PU; ++q) issue_u(q, u + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                mfmas();
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (VAR == 0) {
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        issue_tile(0, 0);
        __syncthreads();
        const bool late = wave >= 4 && nt >= 8;
        for (int kt = 0; kt < nt; ++kt) {
            const bool nxt = kt + 1 < nt;
            if (nxt && !late) issue_tile(kt + 1, (kt + 1) & 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                frags(kt & 1, ks);
                mfmas();
                if (ks == 1 && nxt && late) issue_tile(kt + 1, (kt + 1) & 1);
            }
            __syncthreads();
        }
    } else {
        // VAR 1 "P": DMA pieces of the next tile in the L parts of phases 0, 1, 2;  VAR 2 "P01": phases 0, 1;  VAR 3 "P0": all in phase 0;
        // VAR 6 "Pm": inside the MFMA clusters of phases 0, 1, 2 (one piece after every third MFMA);  VAR 7 "Pn": P without the group offset
        constexpr bool STAGGER = VAR != 7;
        constexpr int NDP = VAR == 2 ? 2 : (VAR == 3 ? 1 : 3);               // phases that carry DMA
        constexpr int PER = (LOADS + NDP - 1) / NDP;
        const int grp = wave >> 2;
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind group 0
        for (int kt = 0; kt < nt; ++kt) {
            const int st = kt & 1;
            const bool nxt = kt + 1 < nt;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                frags(st, ph);
                if (VAR != 6 && nxt && ph < NDP) {
#pragma unroll
                    for (int q = ph * PER; q < (ph + 1) * PER && q < LOADS; ++q) issue_piece(q, kt + 1, st ^ 1);
                }
                if (ph == 3) {
                    // this wave's share of the next tile has landed AND its reads of this tile's last k-step have returned BEFORE the
                    // barrier: the other group is one barrier away from reading the next tile / overwriting this one
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (VAR == 6) {
                    int q = ph * PER;
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                            if (((i * NI + j) % 3) == 1 && nxt && ph < NDP && q < (ph + 1) * PER && q < LOADS) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(q, kt + 1, st ^ 1);
                                __builtin_amdgcn_sched_barrier(0);
                                ++q;
                            }
                        }
                } else {
                    mfmas();
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        if (STAGGER && grp == 0) __builtin_amdgcn_s_barrier();
    }

    // minimal epilogue: fp16 scalar stores when asked (verification); the timing runs keep the accumulators live with a store that never happens
    const int mw = m0 + wm * (MI * 32), nw = n0 + wn * (NI * 32);
    if (store) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    C[(int64_t)(mw + i * 32 + mfma32_row(r, lane)) * N + nw + j * 32 + (lane & 31)] = (half_t)acc[i][j][r];
    } else {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 12345.6789f) C[(int64_t)mw * N + nw + lane] = (half_t)s;
    }
}

template <int BN, int VAR>
static void launch(const half_t* A, const half_t* W, half_t* C, int M, int N, int K, int store, hipStream_t st) {
    constexpr int lds = 2 * (256 + BN) * 128;
    auto kern = gemm_probe<BN, VAR>;
    static bool done = false;
    if (!done) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); done = true; }
    hipLaunchKernelGGL(kern, dim3((M / 256) * (N / BN)), dim3(512), lds, st, A, W, C, M, N, K, store);
}

typedef void (*launch_fn)(const half_t*, const half_t*, half_t*, int, int, int, int, hipStream_t);
struct Variant { const char* name; launch_fn f256, f320; };
static const Variant VARS[] = {
    {"B  shipped loop", launch<256, 0>, launch<320, 0>},
    {"P  staggered, DMA ph 0-2", launch<256, 1>, launch<320, 1>},
    {"P01 DMA ph 0-1", launch<256, 2>, launch<320, 2>},
    {"P0 DMA ph 0", launch<256, 3>, launch<320, 3>},
    {"Pm DMA inside MFMA clusters", launch<256, 6>, launch<320, 6>},
    {"Pn no stagger", launch<256, 7>, launch<320, 7>},
    {"U  unit ring, counted vmcnt", launch<256, 4>, launch<320, 4>},
};
constexpr int NVAR = sizeof(VARS) / sizeof(VARS[0]);

static void fill(std::vector<half_t>& v, unsigned seed, float scale) {
    unsigned s = seed * 2654435761u + 12345u;
    for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (half_t)(((int)(s >> 9) % 2001 - 1000) * (scale / 1000.0f)); }
}

int main(int argc, char** argv) {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    // ---- verification: every variant, both tile widths, against a host fp32 reference (transpose-detecting: random rectangular operands)
    {
        const int M = 512, K = 576;
        for (int BN : {256, 320}) {
            const int N = 2 * BN;
            std::vector<half_t> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
            fill(hA, 1, 1.0f); fill(hW, 2, 1.0f);
            std::vector<float> ref((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    float s = 0.0f;
                    for (int k = 0; k < K; ++k) s += (float)hA[(size_t)m * K + k] * (float)hW[(size_t)n * K + k];
                    ref[(size_t)m * N + n] = s;
                }
            half_t *dA, *dW, *dC;
            CHECK(hipMalloc(&dA, hA.size() * 2)); CHECK(hipMalloc(&dW, hW.size() * 2)); CHECK(hipMalloc(&dC, hC.size() * 2));
            CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
            for (int v = 0; v < NVAR; ++v) {
                double worst = 0.0;
                for (int rep = 0; rep < 20; ++rep) {           // repeated: a race shows as an occasional wrong tile
                    CHECK(hipMemsetAsync(dC, 0, hC.size() * 2, st));
                    (BN == 256 ? VARS[v].f256 : VARS[v].f320)(dA, dW, dC, M, N, K, 1, st);
                    CHECK(hipStreamSynchronize(st));
                    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < hC.size(); ++i) {
                        const double d = std::fabs((double)(float)hC[i] - (double)ref[i]) / (1.0 + std::fabs((double)ref[i]));
                        worst = std::max(worst, d);
                    }
                }
                printf("verify BN=%d %-26s max rel err over 20 runs %.3e %s\n", BN, VARS[v].name, worst, worst < 2e-3 ? "ok" : "FAIL");
            }
            CHECK(hipFree(dA)); CHECK(hipFree(dW)); CHECK(hipFree(dC));
        }
    }
    // ---- timing: interleaved rounds, random operands (guide rule 25), min and median per variant
    struct Shape { const char* name; int M, N, K, BN; };
    const Shape shapes[] = {
        {"L2 ff2    ", 12288, 1280, 5120, 256}, {"L2 ff2/320", 12288, 1280, 5120, 320},
        {"L1 conv-K ", 49152, 640, 5760, 320},  {"L1 ff1    ", 49152, 5120, 640, 256},
        {"L0 conv-K ", 196608, 320, 2880, 320}, {"L1 ff2    ", 49152, 640, 2560, 320},
        {"L2 qkv    ", 12288, 3840, 1280, 320}, {"4096^3    ", 4096, 4096, 4096, 256},
        {"L2 CxC    ", 12288, 1280, 1280, 256}, {"L0 conv1d ", 196608, 320, 960, 320},
    };
    const int rounds = argc > 1 ? atoi(argv[1]) : 7, inner = 6;
    for (int cold = 0; cold < 2; ++cold)
    for (const Shape& s : shapes) {
        const size_t abytes = (size_t)s.M * s.K * 2, wbytes = (size_t)s.N * s.K * 2, cbytes = (size_t)s.M * s.N * 2;
        int nset = 1;
        if (cold) { nset = (int)((size_t)640 * 1024 * 1024 / (abytes + wbytes)) + 1; if (nset < 2) nset = 2; if (nset > 6) nset = 6; }
        std::vector<half_t> hA((size_t)s.M * s.K), hW((size_t)s.N * s.K);
        fill(hA, 3, 1.0f); fill(hW, 4, 0.05f);
        std::vector<half_t*> dA(nset), dW(nset);
        half_t* dC;
        CHECK(hipMalloc(&dC, cbytes));
        for (int i = 0; i < nset; ++i) {
            CHECK(hipMalloc(&dA[i], abytes)); CHECK(hipMalloc(&dW[i], wbytes));
            CHECK(hipMemcpy(dA[i], hA.data(), abytes, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dW[i], hW.data(), wbytes, hipMemcpyHostToDevice));
        }
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        std::vector<std::vector<float>> t(NVAR);
        int rot = 0;
        for (int r = 0; r < rounds + 1; ++r)
            for (int v = 0; v < NVAR; ++v) {
                launch_fn f = s.BN == 256 ? VARS[v].f256 : VARS[v].f320;
                CHECK(hipEventRecord(e0, st));
                for (int i = 0; i < inner; ++i, ++rot) f(dA[rot % nset], dW[rot % nset], dC, s.M, s.N, s.K, 0, st);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) t[v].push_back(ms / inner * 1e3f);
            }
        const double fl = 2.0 * s.M * s.N * s.K;
        printf("%s %s M=%6d N=%5d K=%5d BN=%d:", cold ? "cold" : "warm", s.name, s.M, s.N, s.K, s.BN);
        for (int v = 0; v < NVAR; ++v) {
            std::sort(t[v].begin(), t[v].end());
            const float md = t[v][t[v].size() / 2];
            printf("  [%s] %.1f us %.0f TF", VARS[v].name, md, fl / (md * 1e-6) / 1e12);
        }
        printf("\n");
        fflush(stdout);
        for (int i = 0; i < nset; ++i) { CHECK(hipFree(dA[i])); CHECK(hipFree(dW[i])); }
        CHECK(hipFree(dC));
    }
    return 0;
}
