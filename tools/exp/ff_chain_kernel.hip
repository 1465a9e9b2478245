// EXPERIMENT RECORD, not part of libpanacea_hip.so since round 4 (64 spilled VGPRs and 5 s of every build for a launch
// the product never took).  Round 3 built this fused feed-forward, pinned it green and measured it no faster than the launch
// sequence; the source stays here with its probe (ffchain_probe.hip: hipcc -I ../../include -I ../../panacea_amd/csrc).
// The packing helper engine.pk_ff_chain and the C-ABI entry pnc_ff_chain_f16 are in the history (commit 07bbb9c).
// ff_chain.hip — the feed-forward of a BasicTransformerBlock in ONE launch (gfx950):
//
//     out = x + W2 · geglu(W1 · LN(x) + b1) + b2            (attention.py:91-117, 726-747:  x = ff(norm3(x)) + x)
//
// The unfused form is four launches around three HBM round trips per row: the LayerNorm output (0.64 KB per row at C = 320),
// the 4C GEGLU hidden state (2.5 KB written by the first GEMM, read back by the second) and the fp32 stream — at level 0
// (M = 196 608 rows) 1.7 GB per feed-forward against 0.44 GB of input + output.  Here a row never leaves the chip between
// its fp32 input and its output:
//
//   * EVERYTHING PER TOKEN LIVES IN REGISTERS.  Every contraction is computed transposed — weights are the A operand, tokens the
//     B operand / the columns of C — so a lane owns a token: wave w of the workgroup owns 32 rows, their fp32 stream sits in
//     C/32 accumulator blocks X[b] (C^T layout: lane = token, registers = channels), LayerNorm is a reduction over a lane's
//     registers (+ one xor-32 exchange), the normalised activations are 20 B fragments cut from those registers, the GEGLU product
//     is formed between two accumulator blocks that hold a value and its gate at the same position, and its fp16 result is again
//     a B fragment.  No LDS round trip for any activation.  One wave per SIMD, the whole 512-entry register file (160 + 64
//     accumulator registers, 80 of activation fragments, 32 of weight fragments in flight).
//   * ONLY WEIGHTS MOVE, AS A TAPE.  The host packs W1 | W2 once into the exact order and register image the MFMAs consume
//     (include/panacea_hip.h section 1b): 1 KB per fragment = 64 lanes x 16 bytes.  The kernel streams that tape through a ring of
//     seven 20 KB slots with LDS-DMA (the LDS image of a DMA instruction is lane-linear, i.e. byte-identical to the tape); a
//     fragment read is `slot + 1024 f + 16 lane`: no address arithmetic, no bank conflicts.  The K order inside every aligned
//     group of 16 is permuted on the host (PERM16) so that a C^T accumulator, read register by register, IS the B operand.
//   * Software pipeline over chunks of 32 hidden units: [value | gate rows of chunk c] x LN(x) run while the GEGLU arithmetic of
//     chunk c-1 fills the VALU, then W2's columns of chunk c-1 accumulate onto X.
//
// What the first version taught (tools/exp/ffchain_probe.hip, profiles/round3/ffchain_probe_*.txt: 857 us at M = 196 608, no
// faster than the launch sequence it replaces, with the MFMAs alone at 281 us and the tape DMA alone at 138 us):
//   - row-strided 16 / 8 / 4-byte global accesses of the C^T layout cost 350 us (41 %): rows now pass through a wave-private LDS
//     staging area, whole 128-256 byte row segments per access on the global side, in the prologue and in the epilogue;
//   - the tabulated-Phi GELU put an LDS round trip into every slice of VALU work (118 us exposed): the gate is now pure VALU
//     (Abramowitz-Stegun 7.1.26 erf, |d Phi| < 1e-7, one v_rcp + one v_exp);
//   - five DMA issues at the head of every stage, when the matrix pipe is empty, cost 170-200 us: one piece now follows each
//     group of four MFMAs, where its issue slot is covered.
//
// Bounds at C = 320: 2.46 GFLOP and 2.4 MB of tape per 128 rows; MFMA 37 us per workgroup, 6 workgroups per CU at M = 196 608.
// Built for C = 320 (level 0 of the network, where a workgroup can own whole rows: 21 of the 69 blocks, 40 % of the feed-forward
// time); other widths run the GEMM pair.
#include "gemm_kernel.h"

namespace {

constexpr int FC = 320;                 // channels
constexpr int NB = FC / 32;             // accumulator blocks of the stream
constexpr int NKS = FC / 16;            // k-steps of a contraction over the channels
constexpr int CH = 32;                  // hidden units per chunk
constexpr int STAGE_FR = 20;            // fragments (1 KB each) per tape stage
constexpr int STAGE_BYTES = STAGE_FR * 1024;
constexpr int RING = 7, DEPTH = 5;      // slots; stages in flight ahead of the one being consumed (DEPTH <= RING - 2)
constexpr int ROWS = 128;               // rows per workgroup: 4 waves x 32 tokens
constexpr int MAX_INNER = 1536;         // b1 (2 x inner floats) is kept in LDS: 12 KB
constexpr int B1_BYTES = 2 * MAX_INNER * 4;
constexpr int LDS_BYTES = RING * STAGE_BYTES;         // dynamic: the tape ring (140 KB); static: b1 12 KB
constexpr int GR = 4, NG = STAGE_FR / GR;             // a stage's fragments are consumed in five groups of four
// staging of row segments (64 channels of 32 rows) between the C^T register layout and whole-row global accesses: row pitches
// that put the 16 lanes of a ds_read / ds_write group on 16 different 16-byte slots of the 256-byte bank row
constexpr int P32 = 256 + 16, P16 = 128 + 16, P8 = 64 + 16;
constexpr int E32 = 0, E16 = 32 * P32, E8 = E16 + 32 * P16, EL16 = E8 + 32 * P8, ESTAGE = EL16 + 32 * P16;   // 20 480 B per wave
static_assert(4 * ESTAGE <= LDS_BYTES && 4 * 32 * P32 <= (RING - DEPTH) * STAGE_BYTES, "staging areas live in the ring");

// LDS staging accesses are written as asm: the staging areas live inside the array the tape DMA writes, and hipcc puts a
// `s_waitcnt vmcnt(0)` in front of every compiler-visible LDS read that may alias LDS-DMA memory — which would drain the DMA
// prologue and, in the epilogue, serialise the global stores (vmcnt counts stores on gfx950).
__device__ __forceinline__ void lds_w128(unsigned a, f32x4 v) { asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w64(unsigned a, uint2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_w32(unsigned a, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 lds_r128(unsigned a) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// wait for this wave's LDS operations; the values just read are operands, so that no use of them can move above the wait
__device__ __forceinline__ void lds_wait(f32x4 (&v)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ half8v ldfrag(const char* slot, int f, int lane) {
    return *reinterpret_cast<const half8v*>(slot + f * 1024 + lane * 16);
}

// g * Phi(g), Phi = (1 + erf(g / sqrt 2)) / 2 with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): pure VALU — a table
// lookup here is an LDS round trip in the middle of every slice of GEGLU work
__device__ __forceinline__ float gelu_as(float g) {
    const float x = fabsf(g) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float y = fmaf(t, 1.061405429f, -1.453152027f);
    y = fmaf(y, t, 1.421413741f);
    y = fmaf(y, t, -0.284496736f);
    y = fmaf(y, t, 0.254829592f);
    const float hc = 0.5f * (y * t) * __builtin_amdgcn_exp2f(-(x * x) * 1.44269504088896340736f);     // erfc(x) / 2
    return g * (g >= 0.0f ? 1.0f - hc : hc);
}

// ABL: compile-time ablation mask of tools/exp/ffchain_probe.hip (where does the time go); the library instantiates 0 only.
//   1 no tape DMA in the loop  2 no MFMAs  4 no fragment reads  8 no GEGLU arithmetic  16 no barriers  32 no global loads / stores
template <int ABL>
__global__ __launch_bounds__(256, 1) void ff_chain_kernel(const PncFfChainParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // W1's bias is an LDS variable of its own, filled with ordinary stores (see lds_w128 above: a table inside the DMA'd array
    // would get a vmcnt(0) in front of every read, draining the tape's ring)
    __shared__ __attribute__((aligned(16))) float s_b1[B1_BYTES / 4];
    char* const ring = smem;
    const unsigned ring_lds = (unsigned)(uintptr_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, g = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS + wave * 32;          // first row of this wave
    const int nchunks = p.inner / CH;
    const int nstages = 3 * nchunks;                 // two stages of W1 (value | gate rows interleaved) + one of W2 per chunk
    const char* __restrict__ tape = reinterpret_cast<const char*>(p.tape);

    // piece `grp` of stage i: wave w moves fragment w + 4 grp (five 1 KB DMA instructions per stage and wave)
    auto issue_piece = [&](int i, int grp) {
        const int fr = wave + 4 * grp;
        glds16(reinterpret_cast<const half_t*>(tape + (int64_t)i * STAGE_BYTES + fr * 1024 + lane * 16),
               ring + (i % RING) * STAGE_BYTES + fr * 1024);
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < nstages) {
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) issue_piece(i, grp);
        }
    for (int i = tid * 4; i < 2 * p.inner; i += 1024) *reinterpret_cast<f32x4*>(s_b1 + i) = *reinterpret_cast<const f32x4*>(p.b1 + i);

    // ---- this wave's 32 rows: global memory -> (whole 256-byte row segments) -> LDS -> C^T accumulator layout.
    // Block b, register r of lane (tok, g) <-> channel 32 b + chan(r), chan(r) = (r & 3) + 8 (r >> 2) + 4 g ----
    f32x16 X[NB];
    float sm = 0.0f;
    {
        const unsigned stg = ring_lds + DEPTH * STAGE_BYTES + wave * (32 * P32);       // slots DEPTH.. are free until the loop runs
#pragma unroll
        for (int ps = 0; ps < NB / 2; ++ps) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane, r = idx >> 4, c = idx & 15;
                if constexpr (ABL & 32) v[i] = f32x4{0.5f, -1.0f, 2.0f, 0.25f};
                else v[i] = *reinterpret_cast<const f32x4*>(p.x32 + (row0 + r) * p.ldx + 64 * ps + c * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane, r = idx >> 4, c = idx & 15;
                lds_w128(stg + r * P32 + c * 16, v[i]);
            }
            lds_wait();
            f32x4 w[8];
#pragma unroll
            for (int bq = 0; bq < 8; ++bq) w[bq] = lds_r128(stg + tok * P32 + (32 * (bq >> 2) + 8 * (bq & 3) + 4 * g) * 4);
            lds_wait(w);
#pragma unroll
            for (int bq = 0; bq < 8; ++bq)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    X[2 * ps + (bq >> 2)][4 * (bq & 3) + e] = w[bq][e];
                    sm += w[bq][e];
                }
        }
    }
    // ---- LayerNorm of the row, two-pass in registers (the lane pair (l, l + 32) holds the two halves of a row) ----
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / FC);
    float sq = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = X[b][r] - mean; sq = fmaf(d, d, sq); }
    sq += __shfl_xor(sq, 32, 64);
    const float rs = rsqrtf(sq * (1.0f / FC) + p.ln_eps);
    // B fragments of LN(x): k-step s = 2 b + h covers registers 8 h .. 8 h + 7 of block b (PERM16 on the weight side);
    // + b2 on the stream: the second GEMM accumulates straight onto it
    half8v A[NKS];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 32 * b + 8 * q + 4 * g;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_gamma + col);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_beta + col);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                A[2 * b + (r >> 3)][r & 7] = (half_t)fmaf((X[b][r] - mean) * rs, gm[e], bt[e]);
                X[b][r] += bb[e];
            }
        }

    // ---- the tape ----
    // One pipeline step = one stage: wait until this wave's pieces of stage i have landed (five DMA instructions per stage and
    // wave, the pieces of stages i + 1 .. i + DEPTH - 1 may stay in flight), barrier (publishes every wave's pieces of stage i and
    // retires all reads of the slot recycled next), then five groups of four MFMAs, the piece of stage i + DEPTH that follows
    // each group issued under them.
    int st = 0;
    auto stage_begin = [&]() -> const char* {
        if constexpr (!(ABL & 1)) {
            if (st + DEPTH - 1 < nstages) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        const char* slot = ring + (st % RING) * STAGE_BYTES;
        ++st;
        return slot;
    };
    auto feed = [&](int grp) {           // (st was advanced by stage_begin: the stage in progress is st - 1)
        if constexpr (!(ABL & 1)) {
            if (st - 1 + DEPTH < nstages) issue_piece(st - 1 + DEPTH, grp);
        }
    };
    f32x16 Hv[2], Hg[2];                     // value / gate accumulators of the chunk in flight and of the one before
    half8v Hf[2];                            // GEGLU output of the previous chunk: B fragments of the second GEMM
    // Biases enter as the accumulators' initial value: register r of this lane is hidden unit 32 c + chan(r), four consecutive
    // units per (r >> 2) -> one 16-byte LDS read each (the 32 lanes of a group read the same address: a broadcast)
    auto init_acc = [&](f32x16& acc, const float* bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + 8 * q + 4 * g);
            acc[4 * q] = bb[0]; acc[4 * q + 1] = bb[1]; acc[4 * q + 2] = bb[2]; acc[4 * q + 3] = bb[3];
        }
    };
    // The reads of group k + 1 are issued before the MFMAs of group k (32 registers of fragments in flight: hipcc otherwise hoists
    // a whole stage's reads and spills the activations); a slice of the previous chunk's GEGLU arithmetic and one DMA piece follow
    // each group — work that issues while the matrix pipe drains the four MFMAs just queued.
    half8v wf[2][GR];
    auto rd = [&](const char* slot, int grp, auto b_) {
        constexpr int bb = decltype(b_)::value;
#pragma unroll
        for (int k = 0; k < GR; ++k) {
            if constexpr (!(ABL & 4)) wf[bb][k] = ldfrag(slot, grp * GR + k, lane);
            else wf[bb][k] = A[(grp * GR + k) % NKS];
        }
    };
    // GEGLU of the previous chunk, element r of 16: h = value * gelu(gate) -> B fragment of the second GEMM
    auto geglu1 = [&](const f32x16& v, const f32x16& gt, int r) {
        if constexpr (!(ABL & 8)) Hf[r >> 3][r & 7] = (half_t)(v[r] * gelu_as(gt[r]));
        else Hf[r >> 3][r & 7] = (half_t)(v[r] + gt[r]);
    };
    const std::integral_constant<int, 0> B0{};
    const std::integral_constant<int, 1> B1{};
    // first GEMM, one stage = ten k-steps of BOTH the value and the gate rows of the chunk, fragments interleaved (V_s, G_s): two
    // accumulators alternate, so no MFMA waits for the one issued just before it
    auto gemm1 = [&](f32x16& av, f32x16& ag, const char* slot, int half, const f32x16* pv, const f32x16* pg) {
        rd(slot, 0, B0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) { if (grp & 1) rd(slot, grp + 1, B0); else rd(slot, grp + 1, B1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GR; ++k) {
                const int f = grp * GR + k, s = half * (NKS / 2) + (f >> 1);
                if constexpr (ABL & 2) { asm volatile("" ::"v"(wf[grp & 1][k])); continue; }
                if (f & 1) ag = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][k], A[s], ag, 0, 0, 0);
                else av = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][k], A[s], av, 0, 0, 0);
            }
            feed(grp);
            if (pv) {                        // 8 elements per stage over its five groups: 2 2 2 1 1
                const int e0 = half * 8 + (grp < 3 ? 2 * grp : 3 + grp);
                geglu1(*pv, *pg, e0);
                if (grp < 3) geglu1(*pv, *pg, e0 + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // second GEMM, one stage: X[b] += W2[rows of block b][the previous chunk's 32 hidden units] . h   (fragments in (h, b) order)
    auto gemm2 = [&](const char* slot) {
        rd(slot, 0, B0);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) { if (grp & 1) rd(slot, grp + 1, B0); else rd(slot, grp + 1, B1); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GR; ++k) {
                const int f = grp * GR + k, h = f / NB, b = f - h * NB;
                if constexpr (ABL & 2) { asm volatile("" ::"v"(wf[grp & 1][k])); continue; }
                X[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[grp & 1][k], Hf[h], X[b], 0, 0, 0);
            }
            feed(grp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk = [&](int c, auto cur_, auto prev_) {
        constexpr int cur = decltype(cur_)::value, prev = decltype(prev_)::value;
        // biases of chunk c: value rows at b1[32 c ..], gate rows at b1[inner + 32 c ..]
        init_acc(Hv[cur], s_b1 + c * CH);
        init_acc(Hg[cur], s_b1 + p.inner + c * CH);
        const bool pv = c > 0;
        const char* s0 = stage_begin();
        gemm1(Hv[cur], Hg[cur], s0, 0, pv ? &Hv[prev] : nullptr, &Hg[prev]);    // + GEGLU elements 0..7 of the previous chunk
        const char* s1 = stage_begin();
        gemm1(Hv[cur], Hg[cur], s1, 1, pv ? &Hv[prev] : nullptr, &Hg[prev]);    // + elements 8..15
        if (pv) {
            const char* s2 = stage_begin();
            gemm2(s2);
        }
    };
    const std::integral_constant<int, 0> I0{};
    const std::integral_constant<int, 1> I1{};
    for (int c = 0; c < nchunks; c += 2) {
        chunk(c, I0, I1);
        chunk(c + 1, I1, I0);
    }
    // the last chunk's product and its columns of W2 (nchunks is even: the last chunk used buffer 1)
#pragma unroll
    for (int r = 0; r < 16; ++r) geglu1(Hv[1], Hg[1], r);
    {
        const char* s2 = stage_begin();
        gemm2(s2);
    }

    // ---- outputs: fp32 stream and / or the fp16 operand (+ lo plane) of the next GEMM, C^T registers -> LDS -> whole row
    // segments (64 channels per pass: 256 / 128 / 64 bytes per row) ----
    __builtin_amdgcn_s_barrier();                     // the ring is dead: every wave is past its last fragment read
    const unsigned eb = ring_lds + wave * ESTAGE;
    half_t* o16 = reinterpret_cast<half_t*>(p.out16);
    const bool lo8 = p.out16_lo && p.out_lo_fmt == PNC_LO_E4M3, lo16 = p.out16_lo && p.out_lo_fmt == PNC_LO_F16;
#pragma unroll
    for (int ps = 0; ps < NB / 2; ++ps) {
#pragma unroll
        for (int bq = 0; bq < 8; ++bq) {
            const int b = 2 * ps + (bq >> 2), q = bq & 3, cc = 32 * (bq >> 2) + 8 * q + 4 * g;      // channel inside the pass
            const float v[4] = {X[b][4 * q], X[b][4 * q + 1], X[b][4 * q + 2], X[b][4 * q + 3]};
            if (p.out32) lds_w128(eb + E32 + tok * P32 + cc * 4, f32x4{v[0], v[1], v[2], v[3]});
            if (o16) {
                union { half4v h; uint2 u; } hh;
                hh.h = half4v{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                lds_w64(eb + E16 + tok * P16 + cc * 2, hh.u);
                if (lo8) lds_w32(eb + E8 + tok * P8 + cc, lo_plane4_e4m3(v, hh.h));
                if (lo16) {
                    union { half4v h; uint2 u; } ll;
                    ll.h = lo_plane4(v, hh.h);
                    lds_w64(eb + EL16 + tok * P16 + cc * 2, ll.u);
                }
            }
        }
        lds_wait();
        if constexpr (ABL & 32) continue;
        f32x4 w[8];
        if (p.out32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int idx = i * 64 + lane; w[i] = lds_r128(eb + E32 + (idx >> 4) * P32 + (idx & 15) * 16); }
            lds_wait(w);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = i * 64 + lane;
                *reinterpret_cast<f32x4*>(p.out32 + (row0 + (idx >> 4)) * p.ldo32 + 64 * ps + (idx & 15) * 4) = w[i];
            }
        }
        if (o16) {
            // fp16 plane: 8 lanes per row; e4m3 plane: 4 lanes per row; fp16 lo plane: as the fp16 plane
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int idx = i * 64 + lane; w[i] = lds_r128(eb + E16 + (idx >> 3) * P16 + (idx & 7) * 16); }
#pragma unroll
            for (int i = 0; i < 2; ++i) { const int idx = i * 64 + lane; w[4 + i] = lo8 ? lds_r128(eb + E8 + (idx >> 2) * P8 + (idx & 3) * 16) : w[0]; }
            w[6] = w[0]; w[7] = w[0];
            lds_wait(w);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = i * 64 + lane;
                *reinterpret_cast<f32x4*>(o16 + (row0 + (idx >> 3)) * p.ldo16 + 64 * ps + (idx & 7) * 8) = w[i];
            }
            if (lo8) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int idx = i * 64 + lane;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(p.out16_lo) + (row0 + (idx >> 2)) * p.ldo16 + 64 * ps + (idx & 3) * 16) = w[4 + i];
                }
            }
            if (lo16) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int idx = i * 64 + lane; w[i] = lds_r128(eb + EL16 + (idx >> 3) * P16 + (idx & 7) * 16); }
                lds_wait(w);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = i * 64 + lane;
                    *reinterpret_cast<f32x4*>(reinterpret_cast<half_t*>(p.out16_lo) + (row0 + (idx >> 3)) * p.ldo16 + 64 * ps + (idx & 7) * 8) = w[i];
                }
            }
        }
        lds_wait();                                  // this pass's reads are done before the next pass's writes
    }
}

}  // namespace

extern "C" int pnc_ff_chain_supported(int M, int C, int inner) {
    return (C == FC && M > 0 && M % ROWS == 0 && inner > 0 && inner % 128 == 0 && inner <= MAX_INNER) ? 1 : 0;
}

extern "C" int64_t pnc_ff_chain_tape_bytes(int C, int inner) {
    if (C != FC || inner <= 0 || inner % 128 || inner > MAX_INNER) return 0;
    return (int64_t)3 * (inner / CH) * STAGE_BYTES;
}

extern "C" int pnc_ff_chain_f16(const PncFfChainParams* pp, void* stream) {
    if (!pp) return PNC_EINVAL;
    const PncFfChainParams& p = *pp;
    if (p.struct_bytes != (int32_t)sizeof(PncFfChainParams)) return PNC_EABI;
    if (!p.x32 || !p.tape || !p.ln_gamma || !p.ln_beta || !p.b1 || !p.b2 || (!p.out32 && !p.out16)) return PNC_EINVAL;
    if (!pnc_ff_chain_supported(p.M, p.C, p.inner)) return PNC_EINVAL;
    if (p.out16_lo && (!p.out16 || (p.out_lo_fmt != PNC_LO_F16 && p.out_lo_fmt != PNC_LO_E4M3))) return PNC_EINVAL;
    // whole-row-segment accesses: 16-byte vectors on every plane
    if (p.ldx % 4 || p.ldx < p.C || (p.out32 && (p.ldo32 % 4 || p.ldo32 < p.C)) || (p.out16 && (p.ldo16 % 16 || p.ldo16 < p.C))) return PNC_EALIGN;
    if (((uintptr_t)p.x32 | (uintptr_t)p.tape | (uintptr_t)p.ln_gamma | (uintptr_t)p.ln_beta | (uintptr_t)p.b1 | (uintptr_t)p.b2 |
         (uintptr_t)p.out32 | (uintptr_t)p.out16 | (uintptr_t)p.out16_lo) & 15)
        return PNC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ff_chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL(ff_chain_kernel<0>, dim3(p.M / ROWS), dim3(256), LDS_BYTES, st, p);
    return pnc_launch_status();
}
