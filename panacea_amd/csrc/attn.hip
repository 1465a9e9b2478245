// attn.hip — attention kernels of the decomposed-4D attention (gfx950).
//
// 1. attn_views_kernel: flash-style attention, head dim 64, over width-sliced "views" of a token
//    grid.  Covers intra-view, cross-view (incl. the view-5 one-neighbour quirk) and text
//    cross-attention (include/panacea_hip.h §2).  No tensor is ever re-laid out: a view is a
//    strided window of the (H, W) token grid, addressed directly.
//
//    Everything is computed TRANSPOSED so that one lane owns one query:
//        S^T[key][q] = K Q^T      (A = K tile rows, B = Q fragment kept in registers)
//        O^T[d][q]   = V^T P^T    (A = V^T tile rows, B = P converted in-register)
//    With v_mfma_f32_32x32x16_f16 the C/D column is lane&31, so the softmax statistics (m, l),
//    the probabilities and the output accumulator of query q all live in lane q (and its partner
//    lane q+32, which holds the other half of the keys): the only cross-lane traffic per KV tile is
//    one xor-32 exchange of the running max.  K rows are fed to the MFMA in the permuted order
//    pi(i) = 16*((i>>2)&1) + 4*(i>>3) + (i&3) so that lane group g = lane>>5 ends up holding the 16
//    CONSECUTIVE keys 16g..16g+15 of each 32-key half; P then feeds the second MFMA directly as the
//    B operand and the matching V^T fragment is one 16-byte LDS read (keys contiguous because V is
//    produced channel-major by the projection GEMM's transposed epilogue).
//
// 2. attn_temporal_kernel: self-attention over the T <= 8 frames of one pixel.  25 GFLOP per step
//    in total, so it is a bandwidth-bound VALU kernel: 8 lanes per (token, head), 16-byte loads.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int KT = 64;    // keys per tile; a workgroup of NW waves serves NW*32 queries that share every K/V tile

__device__ __forceinline__ int kperm(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

__device__ __attribute__((aligned(16))) half_t g_attn_zero_chunk[8];   // zero-initialised: DMA source of masked chunks

// NW waves per workgroup, QB blocks of 32 queries per wave: a workgroup serves NW*QB*32 queries that share every
// K/V tile, and with QB = 2 every K / V^T fragment read from LDS feeds two MFMAs (the inner loop is LDS-read bound).
// DMA = true (views whose V^T rows are 16-byte addressable: every level of the network): K / V^T tiles go HBM -> LDS
// with global_load_lds_dwordx4, no VGPR round trip and no ds_write; wv_shift = log2(view width) when it is a power of
// two (every level: 64 / 32 / 16 / 8) so that the per-lane key -> (row, column) split of each tile is a shift, not an
// integer division.  Staging was 25-30 % of the kernel at level 0 (ablation: tools/exp, profiles/round1).
// 4 waves x 2 blocks is compiled for TWO waves per SIMD (<= 256 registers, second launch-bounds argument = waves per EU): two
// such workgroups share a CU without sharing a barrier, so the softmax (VALU) phase of one can run under the MFMA phases of
// the other — the 8-wave workgroup's waves are re-aligned by its barrier every tile (profiles/round2/attn_pmc_l0_intra_r2m.txt:
// VALU 64 % + MFMA 29 % of the SIMD time, one after the other).
template <int NW, int QB, bool DMA>
__global__ __launch_bounds__(64 * NW, (NW == 4 && QB == 2 && DMA) ? 2 : 1) void attn_views_kernel(const PncAttnParams p, const int wv_shift,
                                                                                          const float defer_thr, const int dma_mode,
                                                                                          const float sum_lim) {
    constexpr int QT = NW * QB * 32;
    constexpr int ROWS_PER_IT = NW * 8;          // K / V^T rows staged per iteration (8 rows per wave)
    constexpr int ST_IT = 64 / ROWS_PER_IT;      // 2 (4 waves) or 1 (8 waves)
    constexpr int NST = DMA ? 3 : 2;             // DMA: ring of three stages, two tiles in flight (counted vmcnt)
    __shared__ __attribute__((aligned(16))) char smem[NST * 2 * KT * 128];   // [stage][K | Vt][64 rows x 128 B]
    const half_t* __restrict__ Q = reinterpret_cast<const half_t*>(p.q);
    const half_t* __restrict__ K = reinterpret_cast<const half_t*>(p.k);
    const half_t* __restrict__ VT = reinterpret_cast<const half_t*>(p.vt);
    half_t* __restrict__ O = reinterpret_cast<half_t*>(p.o);
    // halo views (PncAttnParams.k_halo): kv view id -1 / kv_views = view column 0 of a buffer with the band's geometry
    auto seg_view = [&](int id, const half_t*& Kb, const half_t*& Vb) -> int {
        Kb = K; Vb = VT;
        if (id < 0) { Kb = reinterpret_cast<const half_t*>(p.k_halo[0]); Vb = reinterpret_cast<const half_t*>(p.vt_halo[0]); return 0; }
        if (id >= p.kv_views) { Kb = reinterpret_cast<const half_t*>(p.k_halo[1]); Vb = reinterpret_cast<const half_t*>(p.vt_halo[1]); return 0; }
        return id;
    };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: LDS-DMA destinations (M0) and wave-row tests stay on the SALU
    // Workgroups that read the same K / V^T — the query tiles of one (view, frame, head), and the neighbouring views of the cross-view
    // launch — get CONSECUTIVE virtual ids inside one XCD (round 4).  Hardware hands linear workgroup ids to the 8 XCDs round-robin,
    // so as a 3-D grid the 4-8 query tiles of a view landed on 4-8 different XCDs and every one of their L2s fetched that view's
    // keys and values from the fabric for itself (profiles/round4: 26 -> 50 GB per step when the tiles went from 512 to 256 queries).
    const int Wv = p.W / p.views, Nq = p.H * Wv;
    const int nqt = (Nq + QT - 1) / QT;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int qtile = vb % nqt, view = (vb / nqt) % p.views;
    const int gh = vb / (nqt * p.views);
    const int g = gh / p.heads, head = gh % p.heads;
    const int kvWv = p.kvW / p.kv_views;
    const int Nkv = p.kvH * kvWv;                    // keys per kv view
    const bool vec_v = ((kvWv & 7) == 0) && ((p.kvW & 7) == 0) && ((p.ldvt & 7) == 0) && ((p.vt_gstride & 7) == 0);
    const int kvg = g / p.q_per_kv;
    const int hc = head * 64;
    const int grp = lane >> 5;

    // ---- this lane's queries (columns of S^T / O^T): one per query block ----
    bool qok[QB];
    int qloc[QB];                     // view-local query index of this lane (the causal limit)
    int64_t qrow[QB];
    half8v qf[QB][4];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int ql = qtile * QT + (wave * QB + qb) * 32 + (lane & 31);   // view-local query index
        qok[qb] = ql < Nq;
        const int qlc = qok[qb] ? ql : (Nq - 1);
        qloc[qb] = qlc;
        const int qy = qlc / Wv, qx = view * Wv + (qlc - qy * Wv);
        qrow[qb] = ((int64_t)g * p.H + qy) * p.W + qx;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds)
            qf[qb][ds] = *reinterpret_cast<const half8v*>(Q + qrow[qb] * p.ldq + hc + ds * 16 + grp * 8);
    }

    f32x16 oacc[QB][2];
    float mrun[QB], lrun[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        mrun[qb] = -1e30f; lrun[qb] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[qb][0][r] = 0.0f; oacc[qb][1][r] = 0.0f; }
    }
    const float sc = p.scale * 1.44269504088896340736f;   // exp2 domain

    // ---- staging assignment: 16-B chunk column sc8 of rows sr (+ ROWS_PER_IT) ----
    // register path: thread writes LOGICAL chunk sc8 to its swizzled slot.  DMA path: the wave image is lane-linear, so
    // physical slot sc8 of row sr receives logical chunk sc8 ^ ((sr>>1)&7) (ROWS_PER_IT is a multiple of 16)
    const int sc8 = DMA ? ((tid & 7) ^ (((tid >> 3) >> 1) & 7)) : (tid & 7), sr = tid >> 3;
    auto div_wv = [&](int key, int& ky, int& kxl) {
        if (wv_shift >= 0) { ky = key >> wv_shift; kxl = key & ((1 << wv_shift) - 1); }
        else { ky = key / kvWv; kxl = key - ky * kvWv; }
    };
    const int nseg = p.nseg[view];
    const int tiles_per_seg = (Nkv + KT - 1) / KT;
    const int ntiles = nseg * tiles_per_seg;

    half8v rk[ST_IT], rv[ST_IT];
    auto load_tile = [&](int t) {
        const int s = t / tiles_per_seg, tt = t - s * tiles_per_seg;
        const half_t *K, *VT;                      // (this segment's buffers: the band's, or a halo view's)
        const int kview = seg_view(p.seg[view][s], K, VT);
        const int key0 = tt * KT;
        half8v z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < ST_IT; ++i) {
            // K: row = key, chunk = 8 channels
            const int key = key0 + sr + ROWS_PER_IT * i;
            if (key < Nkv) {
                int ky, kxl;
                div_wv(key, ky, kxl);
                const int64_t krow = (int64_t)kvg * p.kv_rows_per_group + (int64_t)ky * p.kvW + kview * kvWv + kxl;
                rk[i] = *reinterpret_cast<const half8v*>(K + krow * p.ldk + hc + sc8 * 8);
            } else rk[i] = z;
            // V^T: row = channel d, chunk = 8 consecutive keys
            const int d = sr + ROWS_PER_IT * i;
            const int kc = key0 + sc8 * 8;
            if (kc < Nkv) {
                const half_t* vrow = VT + (int64_t)kvg * p.vt_gstride + (int64_t)(hc + d) * p.ldvt;
                if (vec_v) {          // 8 consecutive keys of one grid row: one aligned 16-byte load
                    int ky, kxl;
                    div_wv(kc, ky, kxl);
                    rv[i] = *reinterpret_cast<const half8v*>(vrow + (int64_t)ky * p.kvW + kview * kvWv + kxl);
                } else {              // narrow views (< 8 columns or unaligned): gather key by key
                    half8v gth = z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int key2 = kc + e;
                        if (key2 < Nkv) {
                            const int ky = key2 / kvWv, kx = kview * kvWv + (key2 - ky * kvWv);
                            gth[e] = vrow[(int64_t)ky * p.kvW + kx];
                        }
                    }
                    rv[i] = gth;
                }
            } else rv[i] = z;
        }
    };
    auto store_tile = [&](int stage) {
        char* sk = smem + stage * (2 * KT * 128);
        char* sv = sk + KT * 128;
#pragma unroll
        for (int i = 0; i < ST_IT; ++i) {
            *reinterpret_cast<half8v*>(sk + lds_off128(sr + ROWS_PER_IT * i, sc8)) = rk[i];
            *reinterpret_cast<half8v*>(sv + lds_off128(sr + ROWS_PER_IT * i, sc8)) = rv[i];
        }
    };

    // Incremental DMA addresses (round 4): when a tile covers whole rows of the key view (KT % view width == 0: every level of the
    // network) and the view holds whole tiles, a lane's K row / V^T chunk of tile (segment s, tile tt) is a LANE constant plus a
    // WAVE-UNIFORM offset tt * (KT / Wv) * kvW + seg[s] * Wv (in rows of K, in keys of V^T): the per-tile key -> (row, column)
    // split, the bound tests and the 64-bit multiplies (~50 VALU per tile and wave, next to ~280 of softmax) leave the loop;
    // what remains is scalar arithmetic and one 64-bit add per DMA instruction.  Same addresses: bit-identical.
    const bool inc_addr = DMA && dma_mode == 1 && wv_shift >= 0 && kvWv <= KT && (Nkv % KT) == 0 && nseg <= 2;
    const half_t* kptr[ST_IT];
    const half_t* vptr[ST_IT];
#pragma unroll
    for (int i = 0; i < ST_IT; ++i) {
        const int r = sr + ROWS_PER_IT * i;
        const int sh = wv_shift < 0 ? 0 : wv_shift;
        kptr[i] = K + ((int64_t)kvg * p.kv_rows_per_group + (int64_t)(r >> sh) * p.kvW + (r & ((1 << sh) - 1))) * p.ldk + hc + sc8 * 8;
        vptr[i] = VT + (int64_t)kvg * p.vt_gstride + (int64_t)(hc + r) * p.ldvt + (int64_t)((sc8 * 8) >> sh) * p.kvW + ((sc8 * 8) & ((1 << sh) - 1));
    }
    const int rows_per_tile_kvW = (wv_shift >= 0 ? (KT >> wv_shift) : 0) * p.kvW;
    // (the segment's view index is read ONCE: indexed per tile it is a scalar load + lgkmcnt(0) in front of every tile's DMA)
    const half_t *Ks0, *Vs0, *Ks1, *Vs1;
    const int seg_off0 = seg_view(p.seg[view][0], Ks0, Vs0) * kvWv, seg_off1 = seg_view(p.seg[view][nseg > 1 ? 1 : 0], Ks1, Vs1) * kvWv;
    const int64_t kdel0 = Ks0 - K, kdel1 = Ks1 - K, vdel0 = Vs0 - VT, vdel1 = Vs1 - VT;      // element offsets of a segment's buffers (0: the band's)
    auto dma_tile = [&](int t, int stage) {        // DMA path: same addresses, destination = this wave's 8 rows
        char* sk = smem + stage * (2 * KT * 128) + wave * 1024;
        char* sv = sk + KT * 128;
        if (inc_addr) {                            // (uniform)
            const int s1 = t >= tiles_per_seg ? 1 : 0, tt1 = t - s1 * tiles_per_seg;
            const int64_t urow = (int64_t)tt1 * rows_per_tile_kvW + (s1 ? seg_off1 : seg_off0);
            const int64_t ko = urow * p.ldk + (s1 ? kdel1 : kdel0), vo = urow + (s1 ? vdel1 : vdel0);
#pragma unroll
            for (int i = 0; i < ST_IT; ++i) {
                glds16(kptr[i] + ko, sk + i * (ROWS_PER_IT * 128));
                glds16(vptr[i] + vo, sv + i * (ROWS_PER_IT * 128));
            }
            return;
        }
        const int s = t / tiles_per_seg, tt = t - s * tiles_per_seg;
        const half_t *K, *VT;
        const int kview = seg_view(p.seg[view][s], K, VT);
        const int key0 = tt * KT;
#pragma unroll
        for (int i = 0; i < ST_IT; ++i) {
            const int key = key0 + sr + ROWS_PER_IT * i;
            const half_t* ksrc = g_attn_zero_chunk;
            if (key < Nkv) {
                int ky, kxl;
                div_wv(key, ky, kxl);
                const int64_t krow = (int64_t)kvg * p.kv_rows_per_group + (int64_t)ky * p.kvW + kview * kvWv + kxl;
                ksrc = K + krow * p.ldk + hc + sc8 * 8;
            }
            glds16(ksrc, sk + i * (ROWS_PER_IT * 128));
            const int d = sr + ROWS_PER_IT * i;
            const int kc = key0 + sc8 * 8;
            const half_t* vsrc = g_attn_zero_chunk;
            if (kc < Nkv) {
                int ky, kxl;
                div_wv(kc, ky, kxl);
                vsrc = VT + (int64_t)kvg * p.vt_gstride + (int64_t)(hc + d) * p.ldvt + (int64_t)ky * p.kvW + kview * kvWv + kxl;
            }
            glds16(vsrc, sv + i * (ROWS_PER_IT * 128));
        }
    };

    constexpr int LOADS = 2 * ST_IT;             // DMA instructions per thread and tile
    if (DMA) {
        if (ntiles > 0) dma_tile(0, 0);
        if (ntiles > 1) {
            dma_tile(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");     // tile 0 landed, tile 1 may be in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    } else {
        if (ntiles > 0) { load_tile(0); store_tile(0); }
        __syncthreads();
    }

    const int frow = lane & 31;
    const int krow_lds = kperm(frow);
    for (int t = 0; t < ntiles; ++t) {
        const bool more = (t + 1) < ntiles;
        const bool ahead = DMA && (t + 2) < ntiles;
        if (DMA) { if (ahead) dma_tile(t + 2, (t + 2) % 3); }
        else if (more) load_tile(t + 1);
        const char* sk = smem + (DMA ? (t % 3) : (t & 1)) * (2 * KT * 128);
        const char* sv = sk + KT * 128;
        const int tt = t % tiles_per_seg;
        const int key0 = tt * KT;

        // ---- S^T = K Q^T : two 32-key halves; every K fragment feeds the MFMAs of all query blocks ----
        f32x16 s[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kh][r] = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const half8v kf = *reinterpret_cast<const half8v*>(
                    sk + lds_off128(kh * 32 + krow_lds, ds * 2 + grp));
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    s[qb][kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ds], s[qb][kh], 0, 0, 0);
            }
        // lane holds keys key0 + kh*32 + grp*16 + r  (r = 0..15).  Scores stay raw; the softmax scale is folded
        // into the exp2 argument (one fma per score).  Masking only runs on the tile that crosses kv_valid.
        half8v pf[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (key0 + KT > p.kv_valid || p.causal) {
                // keys at or beyond kv_valid are padding: tile-local key index kh*32 + r (compile-time) against ONE per-lane limit.
                // The limit is kept opaque inside the branch: hipcc otherwise speculates the whole index / compare chain (32 v_or +
                // 61 v_cmp per tile) out of it into EVERY tile's instruction stream, where only the last tile of a segment masks.
                // (causal: keys beyond the lane's own query index as well — PncAttnParams.causal, the text tower)
                int lim = p.kv_valid - key0 - grp * 16;
                if (p.causal) lim = min(lim, qloc[qb] + 1 - key0 - grp * 16);
                asm volatile("" : "+v"(lim));
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kh * 32 + r >= lim) s[qb][kh][r] = -1e30f;
            }
            // SUM-TRIGGERED running max (round 6; PNC_OPT_ATTN_SUM_TRIGGER).  The SQ counters of the level-0 launches say the VALU pipe
            // is the busy one (VALU active ~51 cycles per 32-cycle MFMA, profiles/round6/attn_counter_audit.md), and the row maximum is
            // 16 v_max3 (half rate: tools/exp/valu_rate.hip) + a cross-lane exchange per query block and tile — 14 % of the softmax's
            // cycles spent to find out, in every tile after the first, that the running maximum does not move.  So after the first tile
            // the probabilities are formed OPTIMISTICALLY against the running maximum as it is, and the row SUM — needed anyway — tells
            // whether that was safe: a lane's 32 probabilities are each <= their sum, so sum < 2^k bounds every P below 2^k (fp16-safe up
            // to 15; fp16 keeps 11 bits at any magnitude, the accumulators are fp32).  Only when some lane's sum reaches the limit — or
            // is not a number: the comparison is written so that NaN / inf trigger too — the tile is redone the exact way below
            // (maximum, rescale, probabilities against the new maximum).  Same softmax; other roundings of P than the per-tile maximum
            // gives (as PNC_OPT_ATTN_DEFER_MAX already states), so not bit-identical across option values.
            bool exact = !(sum_lim > 0.0f) || t == 0;
            float psum = 0.0f;
            if (!exact) {
#pragma unroll
                for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[qb][kh][r], sc, -mrun[qb]));
                        psum += pv;
                        pf[qb][kh][r >> 3][r & 7] = (half_t)pv;
                    }
                exact = __builtin_amdgcn_ballot_w64(!(psum < sum_lim)) != 0;
            }
            if (exact) {
            float tmax = -1e30f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qb][kh][r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sc;          // scaled (exp2) domain; masked -> -inf-ish
            // running max: rescale the accumulators only when some query of the wave raised its max by more than defer_thr (exp2
            // domain).  Until then the old max stays the reference — P = exp2(s - m_old) <= 2^defer_thr, exact in fp32 and far
            // inside fp16 — and the 64 accumulator multiplies + the alpha exp are skipped: on i.i.d. data some lane of a wave
            // raises its max in most tiles (64 lanes x 1/t each), by more than 8 almost never after the first tile
            // (PNC_OPT_ATTN_DEFER_MAX; the first tile always rescales: mrun starts at -1e30)
            if (__builtin_amdgcn_ballot_w64(tmax > mrun[qb] + defer_thr) != 0) {
                const float mnew = fmaxf(mrun[qb], tmax);
                const float alpha = __builtin_amdgcn_exp2f(mrun[qb] - mnew);
                mrun[qb] = mnew;
                lrun[qb] *= alpha;
#pragma unroll
                for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][dh][r] *= alpha;
            }
            psum = 0.0f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // masked entries: fma(-1e30, sc, -mrun) -> exp2 -> 0 (mrun is finite once any key is valid)
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[qb][kh][r], sc, -mrun[qb]));
                    psum += pv;
                    pf[qb][kh][r >> 3][r & 7] = (half_t)pv;
                }
            }
            lrun[qb] += psum;
        }

        // ---- O^T += V^T P^T : every V^T fragment feeds the MFMAs of all query blocks ----
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const half8v vf = *reinterpret_cast<const half8v*>(
                        sv + lds_off128(dh * 32 + frow, kh * 4 + grp * 2 + ss));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        oacc[qb][dh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[qb][kh][ss], oacc[qb][dh], 0, 0, 0);
                }
        if (DMA) {
            // tile t+1 has landed once at most the LOADS instructions of tile t+2 are outstanding; the raw barrier (no
            // compiler vmcnt(0)) publishes every wave's part of it and retires all reads of the stage recycled next
            if (ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            if (more) store_tile((t + 1) & 1);
            __syncthreads();
        }
    }

    // ---- normalise and store: lane owns its queries, channels d = 32*dh + mfma32_row(r, lane) ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float ltot = lrun[qb] + __shfl_xor(lrun[qb], 32, 64);
        const float inv = ltot > 0.0f ? 1.0f / ltot : 0.0f;
        // The lane pair (l, l+32) holds channels 8*r4 + {0..3} / {4..7} of the same query.  One xor-32 exchange per pair
        // of r4 gives each lane 8 CONSECUTIVE channels (r4 even -> lower lane, r4 odd -> upper lane): 4 stores of 16 B
        // per query block instead of 8 stores of 8 B (32-byte instead of 16-byte contiguous pieces per row).
        half_t* orow = O + qrow[qb] * p.ldo + hc;
        const bool vec16 = ((p.ldo & 7) == 0) && (((uintptr_t)p.o & 15) == 0);
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                half4v he, ho;                                  // this lane's channels of r4 = 2 rp and 2 rp + 1
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    he[q] = (half_t)(oacc[qb][dh][(2 * rp) * 4 + q] * inv);
                    ho[q] = (half_t)(oacc[qb][dh][(2 * rp + 1) * 4 + q] * inv);
                }
                if (vec16) {
                    union { half4v h; int2 i; } snd, rcv;
                    snd.h = grp ? he : ho;                      // what the partner lane stores
                    rcv.i.x = __shfl_xor(snd.i.x, 32, 64);
                    rcv.i.y = __shfl_xor(snd.i.y, 32, 64);
                    half8v o8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o8[q] = grp ? rcv.h[q] : he[q];
                        o8[4 + q] = grp ? ho[q] : rcv.h[q];
                    }
                    if (qok[qb]) *reinterpret_cast<half8v*>(orow + dh * 32 + 8 * (2 * rp + grp)) = o8;
                } else if (qok[qb]) {
                    *reinterpret_cast<half4v*>(orow + dh * 32 + 8 * (2 * rp) + 4 * grp) = he;
                    *reinterpret_cast<half4v*>(orow + dh * 32 + 8 * (2 * rp + 1) + 4 * grp) = ho;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// attn_text_kernel (round 6): cross-attention against FEW keys shared by every query of a sample — the 77 text tokens (attention.py:
// 229-291 with the prompt as context; 63 sites per network evaluation).  On attn_views_kernel this launch is a stream of q in / o out
// with a 2-tile loop in the middle: 2.4-3.1 TB/s, because a workgroup's life is one serial chain — wait for its Q rows, two key tiles
// of which the second holds 13 keys, store — and 8 waves per CU cannot hide it (profiles/round6/attn_counter_audit.md: 52 % of the
// wave cycles waiting).  Here a workgroup of 4 waves owns 128 queries x HG heads:
//   * the Q fragments of ALL its heads are requested up front (HG x 64 B per lane: 80 KB per workgroup in flight);
//   * the <= 96 keys are ONE tile, so the softmax is single-pass — true row maximum, no running state, no rescale —, the padded
//     fourth 32-key block of the 2 x 64-key form is never computed (24 MFMAs per 32 queries and head instead of 32, 48 score slots per
//     lane instead of 64);
//   * K / V^T of head h + 1 arrive by LDS-DMA under head h's arithmetic (two stages of 28 KB).
// Same S^T = K Q^T / O^T = V^T P^T register layout as attn_views_kernel (a lane owns a query; K rows permuted by kperm()).
template <int HG>
__global__ __launch_bounds__(256, 2) void attn_text_kernel(const PncAttnParams p, const int nhg) {
    constexpr int KROWS = 96, STAGE = KROWS * 128 + 2 * 64 * 128;      // K tile (96 keys) + two V^T tiles (keys 0-63, 64-127)
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const half_t* __restrict__ Q = reinterpret_cast<const half_t*>(p.q);
    const half_t* __restrict__ K = reinterpret_cast<const half_t*>(p.k);
    const half_t* __restrict__ VT = reinterpret_cast<const half_t*>(p.vt);
    half_t* __restrict__ O = reinterpret_cast<half_t*>(p.o);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nq = p.H * p.W, nqt = (Nq + 127) >> 7;
    const int Nkv = p.kvH * p.kvW;                          // rows of the key buffer per sample (80), kv_valid of them real
    const int item = blockIdx.x;
    const int hg = item % nhg, qt = (item / nhg) % nqt, g = item / (nhg * nqt);
    const int kvg = g / p.q_per_kv;
    const int h0 = hg * HG, nh = min(HG, p.heads - h0);
    const int grp = lane >> 5, frow = lane & 31;
    const int ql = qt * 128 + wave * 32 + frow;
    const bool qok = ql < Nq;
    const int64_t qrow = (int64_t)g * Nq + (qok ? ql : Nq - 1);
    // ---- every head's Q fragments, requested at once
    half8v qf[HG][4];
#pragma unroll
    for (int hh = 0; hh < HG; ++hh)
        if (hh < nh) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds)
                qf[hh][ds] = *reinterpret_cast<const half8v*>(Q + qrow * p.ldq + (h0 + hh) * 64 + ds * 16 + grp * 8);
        }
    // ---- K / V^T of one head -> stage: rows of 128 B, chunk index XOR-swizzled on the source side (as attn_views_kernel's DMA path)
    const int sr = tid >> 3, sc8 = (tid & 7) ^ ((sr >> 1) & 7);
    auto dma_head = [&](int hh, int stage) {
        const int hc = (h0 + hh) * 64;
        char* sk = smem + stage * STAGE + wave * 1024;
#pragma unroll
        for (int i = 0; i < KROWS / 32; ++i) {             // K: row = key, chunk = 8 channels
            const int key = sr + 32 * i;
            const half_t* src = key < Nkv ? K + ((int64_t)kvg * p.kv_rows_per_group + key) * p.ldk + hc + sc8 * 8 : g_attn_zero_chunk;
            glds16(src, sk + i * 4096);
        }
        char* sv = smem + stage * STAGE + KROWS * 128 + wave * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)                         // V^T: row = channel d, chunk = 8 consecutive keys of tile j
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int d = sr + 32 * i, kc = j * 64 + sc8 * 8;
                const half_t* src = kc < Nkv ? VT + (int64_t)kvg * p.vt_gstride + (int64_t)(hc + d) * p.ldvt + kc : g_attn_zero_chunk;
                glds16(src, sv + j * 8192 + i * 4096);
            }
    };
    dma_head(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nh > 1) dma_head(1, 1);
    const float sc = p.scale * 1.44269504088896340736f;
    const int krow_lds = kperm(frow);
    // Padding keys (kv_valid .. 95; their K rows are zeros) are masked through the ACCUMULATOR'S INITIAL VALUE: block 2's first MFMA
    // starts from -1e30 at this lane's padding positions (key 64 + grp * 16 + r >= kv_valid) and from 0 elsewhere — sixteen registers
    // set once per workgroup, no compare / select chain per head (the host dispatches this kernel for 64 < kv_valid <= 96 only)
    f32x16 init2;
#pragma unroll
    for (int r = 0; r < 16; ++r) init2[r] = (64 + grp * 16 + r >= p.kv_valid) ? -1e30f : 0.0f;
    const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const bool vec16 = ((p.ldo & 7) == 0) && (((uintptr_t)p.o & 15) == 0);
#pragma unroll
    for (int hh = 0; hh < HG; ++hh) {
        if (hh >= nh) break;
        const char* sk = smem + (hh & 1) * STAGE;
        const char* sv = sk + KROWS * 128;
        // ---- S^T = K Q^T: three 32-key blocks
        f32x16 s[3];
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const half8v kf = *reinterpret_cast<const half8v*>(sk + lds_off128(kb * 32 + krow_lds, ds * 2 + grp));
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[hh][ds], ds == 0 ? (kb == 2 ? init2 : zero16) : s[kb], 0, 0, 0);
            }
        }
        // ---- single-pass softmax: the keys are one tile
        float tmax = -1e30f;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[kb][r]);
        const float m = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * sc;
        half8v pf[3][2];
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, -m));      // padding: exp2(-huge) = 0
                psum += pv;
                pf[kb][r >> 3][r & 7] = (half_t)pv;
            }
        const float ltot = psum + __shfl_xor(psum, 32, 64);
        // ---- O^T = V^T P^T
        f32x16 oacc[2];
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dh][r] = 0.0f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int kb = 0; kb < 3; ++kb) {
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const half8v vf = *reinterpret_cast<const half8v*>(sv + (kb >> 1) * 8192 + lds_off128(dh * 32 + frow, (kb & 1) * 4 + grp * 2 + ss));
                    oacc[dh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kb][ss], oacc[dh], 0, 0, 0);
                }
            }
        // the next head's operands have landed and every wave is past its reads of this stage
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- normalise and store (attn_views_kernel's epilogue: one xor-32 exchange per pair of r4 -> 16-byte stores)
        const float inv = ltot > 0.0f ? 1.0f / ltot : 0.0f;
        half_t* orow = O + qrow * p.ldo + (h0 + hh) * 64;
        // (16-byte pieces through one xor-32 exchange per pair of r4, as attn_views_kernel: 8-byte stores straight from the accumulator
        // layout save 40 instructions per head and measured 5-9 % SLOWER — the launch is sensitive to store width)
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                half4v he, ho;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    he[q] = (half_t)(oacc[dh][(2 * rp) * 4 + q] * inv);
                    ho[q] = (half_t)(oacc[dh][(2 * rp + 1) * 4 + q] * inv);
                }
                if (vec16) {
                    union { half4v h; int2 i; } snd, rcv;
                    snd.h = grp ? he : ho;
                    rcv.i.x = __shfl_xor(snd.i.x, 32, 64);
                    rcv.i.y = __shfl_xor(snd.i.y, 32, 64);
                    half8v o8;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o8[q] = grp ? rcv.h[q] : he[q];
                        o8[4 + q] = grp ? ho[q] : rcv.h[q];
                    }
                    if (qok) *reinterpret_cast<half8v*>(orow + dh * 32 + 8 * (2 * rp + grp)) = o8;
                } else if (qok) {
                    *reinterpret_cast<half4v*>(orow + dh * 32 + 8 * (2 * rp) + 4 * grp) = he;
                    *reinterpret_cast<half4v*>(orow + dh * 32 + 8 * (2 * rp + 1) + 4 * grp) = ho;
                }
            }
        if (hh + 2 < nh) dma_head(hh + 2, hh & 1);
    }
}

// ------------------------------------------------------------------------------------------
// temporal attention: lane = slot*8 + dl ; slot -> (pixel sub-index, frame t) ; dl -> 8 channels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8(const half8v a, const half8v b) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = fmaf((float)a[i], (float)b[i], s);
    return s;
}

__global__ __launch_bounds__(256) void attn_temporal_kernel(
    const half_t* __restrict__ Q, int ldq, const half_t* __restrict__ Kp, int ldk,
    const half_t* __restrict__ Vp, int ldv, half_t* __restrict__ O, int ldo,
    int B, int T, int Npix, int heads, float scale, int ppw, int64_t nwork) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave work item
    if (w >= nwork) return;
    // work item -> (b, pixel group, head); head fastest so neighbouring waves share DRAM pages
    const int head = (int)(w % heads);
    const int64_t pg = w / heads;
    const int groups_per_b = (Npix + ppw - 1) / ppw;
    const int b = (int)(pg / groups_per_b);
    const int p0 = (int)(pg % groups_per_b) * ppw;
    const int slot = lane >> 3, dl = lane & 7;
    const int ps = slot / T, t = slot - ps * T;
    const int pix = p0 + ps;
    const bool ok = (ps < ppw) && (pix < Npix);
    const int pixc = ok ? pix : p0;
    const int tc = ok ? t : 0;
    const int col = head * 64 + dl * 8;
    const int64_t row_q = ((int64_t)(b * T + tc)) * Npix + pixc;
    const half8v q = *reinterpret_cast<const half8v*>(Q + row_q * ldq + col);
    // Each lane loads the K and V chunk of ITS OWN (pixel, frame) once and parks them in a wave-private LDS slab; the
    // T x T products then read the other frames' chunks from LDS (same-address reads broadcast).  Loading every key /
    // value row once per QUERY frame instead (8x the L1 traffic) held the kernel at 2.4 TB/s of HBM-equivalent.
    __shared__ __attribute__((aligned(16))) half8v skv[4][2][64];
    const int wv = threadIdx.x >> 6;
    skv[wv][0][lane] = *reinterpret_cast<const half8v*>(Kp + row_q * ldk + col);
    skv[wv][1][lane] = *reinterpret_cast<const half8v*>(Vp + row_q * ldv + col);
    // (one wave = one LDS slab: wave-level execution order makes the writes visible to the reads below)
    __builtin_amdgcn_wave_barrier();
    float sc[8];
    float mx = -1e30f;
    const float c = scale * 1.44269504088896340736f;
    const int lbase = (ps * T) * 8 + dl;          // lane of (this pixel, frame 0, this channel chunk)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < T) {
            const half8v k = skv[wv][0][ok ? lbase + s * 8 : lane];
            float d = dot8(q, k);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d += __shfl_xor(d, 4, 64);
            sc[s] = d * c;
            mx = fmaxf(mx, sc[s]);
        } else sc[s] = -1e30f;
    }
    float l = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) { sc[s] = (s < T) ? exp2f(sc[s] - mx) : 0.0f; l += sc[s]; }
    const float inv = 1.0f / l;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < T) {
            const half8v v = skv[wv][1][ok ? lbase + s * 8 : lane];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(sc[s], (float)v[i], acc[i]);
        }
    }
    if (ok) {
        half8v o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (half_t)(acc[i] * inv);
        *reinterpret_cast<half8v*>(O + row_q * ldo + col) = o;
    }
}

}  // namespace

extern "C" int pnc_attn_views_f16(const PncAttnParams* pp, void* stream) {
    if (!pp) return PNC_EINVAL;
    const PncAttnParams& p = *pp;
    if (!p.q || !p.k || !p.vt || !p.o) return PNC_EINVAL;
    if (p.views < 1 || p.views > 8 || p.kv_views < 1 || p.kv_views > 8) return PNC_EINVAL;
    if (p.W % p.views || p.kvW % p.kv_views) return PNC_EINVAL;
    const int kvWv = p.kvW / p.kv_views;
    (void)kvWv;   // any view width: V^T chunks fall back to a per-key gather when a view row is not 8-aligned
    if (p.ldq % 8 || p.ldk % 8 || p.ldo % 4) return PNC_EALIGN;
    if (((uintptr_t)p.q | (uintptr_t)p.k) & 15) return PNC_EALIGN;
    if ((uintptr_t)p.vt & 15) return PNC_EALIGN;
    if ((uintptr_t)p.o & 7) return PNC_EALIGN;
    if (p.q_per_kv < 1 || p.groups < 1 || p.heads < 1) return PNC_EINVAL;
    if (p.kv_valid < 1 || p.kv_valid > p.kvH * kvWv) return PNC_EINVAL;
    if (p.causal != 0 && p.causal != 1) return PNC_EINVAL;
    for (int v = 0; v < p.views; ++v) {
        if (p.nseg[v] < 1 || p.nseg[v] > 2) return PNC_EINVAL;
        for (int s = 0; s < p.nseg[v]; ++s) {
            const int id = p.seg[v][s];
            if (id == -1 || id == p.kv_views) {          // a halo view: its buffers must be there, laid out like the band's
                const int side = id < 0 ? 0 : 1;
                if (!p.k_halo[side] || !p.vt_halo[side]) return PNC_EINVAL;
                if (((uintptr_t)p.k_halo[side] | (uintptr_t)p.vt_halo[side]) & 15) return PNC_EALIGN;
            } else if (id < 0 || id > p.kv_views) {
                return PNC_EINVAL;
            }
        }
    }
    const int Nq = p.H * (p.W / p.views);
    // variants: (waves, query blocks per wave) -> queries per workgroup.  Two blocks per wave: every K / V^T fragment read feeds
    // two MFMAs (600-670 TFLOP/s at level 0 vs 470-500 for 4 x 1); mid-size views: 8 x 1 (256 queries); small views: 4 x 1
    // (128).  Measured in profiles/round1/kbench_attn_variants.log, profiles/round4/attn_ab_r4a.log.
    const int force = pnc_get_option(PNC_OPT_ATTN_VARIANT);      // tests / kbench: force one variant
    // Few keys per view (the 77 text tokens: two K/V tiles): the launch is a stream of q in / o out.  Round 3 took 4 waves x 1 block
    // (three small workgroups per CU: 90 / 48 / 30 us at levels 0-2 vs 100 / 53 / 37 for 8 x 2); with two workgroups per CU the
    // 4 x 2 shape (256 queries per workgroup: half the per-workgroup prologues) is faster still — 84 / 45 / 27 us vs 110 / 55 / 33
    // on one box (tools/exp/text_attn_variants.py, profiles/round4/attn_text_variants_r4k.txt).
    const int kv_keys = p.kv_valid * 2;                           // at most two K/V segments per view
    // Large views (>= 512 queries): 4 waves x 2 blocks, TWO workgroups per CU (round 4).  The 8 x 2 workgroup it replaces shares
    // each K/V tile among 512 queries but its barrier re-aligns the two waves of every SIMD each tile, so their softmax (VALU)
    // and MFMA phases coincide and add up; two independent workgroups drift apart and overlap them: level 0 intra 685 -> 638 us,
    // cross 1221 -> 1164, level 1 intra 111 -> 100 (same box, interleaved: profiles/round4/attn_ab_r4a.log).  force = 1: the size
    // heuristic with 82 in the place of 42 (whole-step A/B of the old choice).
    // Few keys shared by all queries of a sample (the text tokens): the dedicated single-pass kernel (round 6).  PNC_OPT_ATTN_VARIANT = 43
    // forces it wherever it applies (tests), any other non-zero value keeps attn_views_kernel (A/B).
    {
        const bool text_ok = p.views == 1 && p.kv_views == 1 && p.nseg[0] == 1 && p.seg[0][0] == 0 && !p.causal && p.kvH * p.kvW <= 96 &&
                             p.kv_valid > 64 && p.kv_valid <= 96 && (p.ldvt & 7) == 0 && (p.vt_gstride & 7) == 0 && (p.ldk & 7) == 0;
        constexpr int HG = 5;
        const int nhg = (p.heads + HG - 1) / HG;
        const long nwg = (long)((Nq + 127) / 128) * nhg * p.groups;
        // Small per-frame grids — the 4 x 48 level: 2 query tiles x 4 head groups — stay on the smaller workgroups of attn_views_kernel.
        // The test looks at ONE group (frame), never at the batch: a sample's eps must not depend on the batch it is evaluated in
        // (tests/test_model_gpu.py: the CFG half alone reproduces its bits), and the two kernels round P differently.
        const bool big_enough = (long)((Nq + 127) / 128) * nhg >= 16;
        const int dopt = pnc_get_option(PNC_OPT_ATTN_DMA);          // bit 2: keep attn_views_kernel for these launches (whole-step A/B)
        if (text_ok && (force == 43 || (force == 0 && big_enough)) && (dopt & 3) != 0 && !(dopt & 4)) {
            const dim3 grid((unsigned)nwg);
            hipLaunchKernelGGL((attn_text_kernel<HG>), grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, nhg);
            return pnc_launch_status();
        }
    }
    const int big = force == 1 ? 82 : 42;
    const int variant = force >= 41 ? force : (kv_keys <= 256 ? (Nq >= 256 ? 42 : 41) : (Nq >= 512 ? big : (Nq >= 256 ? 81 : 41)));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int dma_mode = pnc_get_option(PNC_OPT_ATTN_DMA) & 3;
    const bool dma = ((kvWv & 7) == 0) && ((p.kvW & 7) == 0) && ((p.ldvt & 7) == 0) && ((p.vt_gstride & 7) == 0) && dma_mode != 0;
    int wv_shift = -1;
    if (kvWv > 0 && (kvWv & (kvWv - 1)) == 0) { wv_shift = 0; while ((1 << wv_shift) < kvWv) ++wv_shift; }
    int dopt = pnc_get_option(PNC_OPT_ATTN_DEFER_MAX);
    const float defer_thr = (float)(dopt < 0 ? 0 : (dopt > 14 ? 14 : dopt));
    const int sopt = pnc_get_option(PNC_OPT_ATTN_SUM_TRIGGER);
    const float sum_lim = sopt <= 0 ? 0.0f : ldexpf(1.0f, sopt > 14 ? 14 : sopt);
#define PNC_ATTN_LAUNCH(NW_, QB_, QTILE_)                                                                         \
    do {                                                                                                          \
        dim3 grid(((Nq + (QTILE_) - 1) / (QTILE_)) * p.views * p.groups * p.heads);                               \
        if (dma) hipLaunchKernelGGL((attn_views_kernel<NW_, QB_, true>), grid, dim3(64 * NW_), 0, st, p, wv_shift, defer_thr, dma_mode, sum_lim);  \
        else hipLaunchKernelGGL((attn_views_kernel<NW_, QB_, false>), grid, dim3(64 * NW_), 0, st, p, wv_shift, defer_thr, dma_mode, sum_lim);   \
    } while (0)
    if (variant == 42) PNC_ATTN_LAUNCH(4, 2, 256);
    else if (variant == 82) PNC_ATTN_LAUNCH(8, 2, 512);
    else if (variant == 81) PNC_ATTN_LAUNCH(8, 1, 256);
    else PNC_ATTN_LAUNCH(4, 1, 128);
#undef PNC_ATTN_LAUNCH
    return pnc_launch_status();
}

extern "C" int pnc_attn_temporal_f16(const void* q, int ldq, const void* k, int ldk,
                                     const void* v, int ldv, void* o, int ldo,
                                     int B, int T, int Npix, int heads, float scale, void* stream) {
    if (!q || !k || !v || !o) return PNC_EINVAL;
    if (T < 1 || T > 8 || B < 1 || Npix < 1 || heads < 1) return PNC_EINVAL;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return PNC_EALIGN;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return PNC_EALIGN;
    const int ppw = 8 / T;
    const int64_t nwork = (int64_t)B * ((Npix + ppw - 1) / ppw) * heads;
    const int64_t blocks = (nwork + 3) / 4;
    hipLaunchKernelGGL(attn_temporal_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const half_t*>(q), ldq, reinterpret_cast<const half_t*>(k), ldk,
                       reinterpret_cast<const half_t*>(v), ldv, reinterpret_cast<half_t*>(o), ldo,
                       B, T, Npix, heads, scale, ppw, nwork);
    return pnc_launch_status();
}
