// ff_chain.hip — the per-token tail of a BasicTransformerBlock in ONE launch (gfx950):
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
//     C/10 accumulator blocks X[b] (C^T layout: lane = token, registers = channels), LayerNorm is a reduction over a lane's
//     registers (+ one xor-32 exchange), the normalised activations are 20 B fragments cut from those registers, the GEGLU product
//     is formed between two accumulator blocks that hold a value and its gate at the same position, and its fp16 result is again
//     a B fragment.  No LDS round trip for any activation.  One wave per SIMD, the whole 512-entry register file (160 accumulator
//     + 80 fragment + 64 hidden-state registers, ...).
//   * ONLY WEIGHTS MOVE, AS A TAPE.  The host packs W1 | W2 once into the exact order and register image the MFMAs consume
//     (pnc_ff_chain layout below): 1 KB per fragment = 64 lanes x 16 bytes.  The kernel streams that tape through a ring of
//     six 20 KB slots with LDS-DMA (the LDS image of a DMA instruction is lane-linear, i.e. byte-identical to the tape); a
//     fragment read is `slot + 1024 f + 16 lane`: no address arithmetic, no bank conflicts.  The K order inside every aligned
//     group of 16 is permuted on the host (PERM16) so that a C^T accumulator, read register by register, IS the B operand.
//   * Software pipeline over chunks of 32 hidden units: [value rows | gate rows of chunk c] x LN(x) run while the GEGLU
//     arithmetic of chunk c-1 (tabulated Phi, as in the GEMM epilogue) fills the VALU, then W2's columns of chunk c-1 accumulate
//     onto X.
//
// Bounds at C = 320: 2.46 GFLOP and 2.4 MB of tape per 128 rows; MFMA 37 us, tape 40-60 us (L2 -> LDS at 40-60 GB/s per CU), HBM 3 KB per
// row.  Built for C = 320 (level 0 of the network, where a workgroup can own whole rows: 21 of the 69 blocks, 40 % of the
// feed-forward time); other widths run the GEMM pair.
#include "gemm_kernel.h"

namespace {

using pnc_gemm::gelu_tab_f;
using pnc_gemm::PHI_BYTES;

constexpr int FC = 320;                 // channels
constexpr int NB = FC / 32;             // accumulator blocks of the stream
constexpr int NKS = FC / 16;            // k-steps of a contraction over the channels
constexpr int CH = 32;                  // hidden units per chunk
constexpr int STAGE_FR = 20;            // fragments (1 KB each) per tape stage
constexpr int STAGE_BYTES = STAGE_FR * 1024;
constexpr int RING = 6, DEPTH = 4;      // slots; stages in flight ahead of the one being consumed (DEPTH <= RING - 2)
constexpr int ROWS = 128;               // rows per workgroup: 4 waves x 32 tokens
constexpr int MAX_INNER = 1536;         // b1 (2 x inner floats) is kept in LDS: 12 KB
constexpr int B1_BYTES = 2 * MAX_INNER * 4;
constexpr int LDS_BYTES = RING * STAGE_BYTES;         // dynamic: the tape ring (120 KB); static: Phi table 16 KB + b1 12 KB

__device__ __forceinline__ half8v ldfrag(const char* slot, int f, int lane) {
    return *reinterpret_cast<const half8v*>(slot + f * 1024 + lane * 16);
}

// ABL: compile-time ablation mask of tools/exp/ffchain_probe.hip (where does the time go); the library instantiates 0 only.
//   1 no tape DMA in the loop  2 no MFMAs  4 no fragment reads  8 no GEGLU arithmetic  16 no barriers  32 no prologue loads / stores
template <int ABL>
__global__ __launch_bounds__(256, 1) void ff_chain_kernel(const PncFfChainParams p, const float* __restrict__ phi_g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The two lookup tables are LDS variables of their own, filled with ordinary stores: hipcc puts `s_waitcnt vmcnt(0)` in front
    // of an LDS read that may alias memory written by LDS-DMA — with the tables inside the DMA'd array that wait sat in front of
    // every GELU lookup and drained the tape's ring (first builds of this kernel).
    __shared__ __attribute__((aligned(16))) float s_phi[PHI_BYTES / 4];
    __shared__ __attribute__((aligned(16))) float s_b1[B1_BYTES / 4];
    char* const ring = smem;
    const float* const phi = s_phi;
    const float* const sb1 = s_b1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, g = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * ROWS + wave * 32 + tok;
    const int nchunks = p.inner / CH;
    const int nstages = 3 * nchunks;                 // two stages of W1 (value | gate rows interleaved) + one of W2 per chunk
    const char* __restrict__ tape = reinterpret_cast<const char*>(p.tape);

    // stage i of the tape -> ring slot i % RING; wave w moves fragments w, w+4, ... (5 DMA instructions per stage)
    auto issue = [&](int i) {
        const char* src = tape + (int64_t)i * STAGE_BYTES + lane * 16;
        char* dst = ring + (i % RING) * STAGE_BYTES;
#pragma unroll
        for (int f = 0; f < STAGE_FR / 4; ++f) {
            const int fr = wave + 4 * f;
            glds16(reinterpret_cast<const half_t*>(src + fr * 1024), dst + fr * 1024);
        }
    };
    // tables: Phi of the GELU (2048 x {Phi, dPhi}) and W1's bias (2 inner floats), 16 bytes per thread and pass
    for (int i = tid * 4; i < PHI_BYTES / 4; i += 1024) *reinterpret_cast<f32x4*>(s_phi + i) = *reinterpret_cast<const f32x4*>(phi_g + i);
    for (int i = tid * 4; i < 2 * p.inner; i += 1024) *reinterpret_cast<f32x4*>(s_b1 + i) = *reinterpret_cast<const f32x4*>(p.b1 + i);
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        if (i < nstages) issue(i);

    // ---- this lane's token: the fp32 stream in C^T accumulator layout.  Block b, register r <-> channel 32 b + chan(r),
    // chan(r) = (r & 3) + 8 (r >> 2) + 4 g: four consecutive channels per (b, r >> 2) -> 16-byte loads ----
    f32x16 X[NB];
    const float* xrow = p.x32 + row * p.ldx;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {1.0f, 2.0f, 3.0f, 4.0f};
            if constexpr (!(ABL & 32)) v = *reinterpret_cast<const f32x4*>(xrow + 32 * b + 8 * q + 4 * g);
            X[b][4 * q] = v[0]; X[b][4 * q + 1] = v[1]; X[b][4 * q + 2] = v[2]; X[b][4 * q + 3] = v[3];
        }
    // ---- LayerNorm of the row, two-pass in registers (the lane pair (l, l + 32) holds the two halves of a row) ----
    float sm = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) sm += X[b][r];
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / FC);
    float sq = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = X[b][r] - mean; sq = fmaf(d, d, sq); }
    sq += __shfl_xor(sq, 32, 64);
    const float rs = rsqrtf(sq * (1.0f / FC) + p.ln_eps);
    // B fragments of LN(x): k-step s = 2 b + h covers registers 8 h .. 8 h + 7 of block b (PERM16 on the weight side)
    half8v A[NKS];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.ln_gamma + 32 * b + 8 * q + 4 * g);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(p.ln_beta + 32 * b + 8 * q + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * q + e;
                A[2 * b + (r >> 3)][r & 7] = (half_t)fmaf((X[b][r] - mean) * rs, gm[e], bt[e]);
            }
        }
    // + b2: the second GEMM accumulates straight onto the stream
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(p.b2 + 32 * b + 8 * q + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) X[b][4 * q + e] += bb[e];
        }

    // ---- the tape ----
    // One pipeline step = one stage: issue stage i + DEPTH, wait until this wave's pieces of stage i have landed (5 DMA
    // instructions per stage and wave: DEPTH stages may stay in flight), barrier (publishes every wave's pieces of stage i and
    // retires all reads of the slot recycled next), MFMAs of stage i.
    int st = 0;
    auto stage_begin = [&]() -> const char* {
        if constexpr (!(ABL & 1)) {
            if (st + DEPTH < nstages) {
                issue(st + DEPTH);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH * (STAGE_FR / 4)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        const char* slot = ring + (st % RING) * STAGE_BYTES;
        ++st;
        return slot;
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
    // A stage's 20 fragments are consumed in four groups of five: the reads of group k + 1 are issued before the MFMAs of group k
    // (40 registers of fragments in flight, not 80: hipcc otherwise hoists a whole stage's reads and spills the activations),
    // and a slice of the previous chunk's GEGLU arithmetic follows each group — VALU work that issues while the matrix pipe
    // drains the five MFMAs just queued.
    constexpr int GR = 5, NG = STAGE_FR / GR;
    half8v wf[2][GR];
    auto rd = [&](const char* slot, int grp, auto b_) {
        constexpr int bb = decltype(b_)::value;
#pragma unroll
        for (int k = 0; k < GR; ++k) {
            if constexpr (!(ABL & 4)) wf[bb][k] = ldfrag(slot, grp * GR + k, lane);
            else wf[bb][k] = A[(grp * GR + k) % NKS];
        }
    };
    // GEGLU of the previous chunk, elements [2 part, 2 part + 2) of 16: h = value * gelu(gate) -> B fragment of the second GEMM
    auto geglu2 = [&](const f32x16& v, const f32x16& gt, int part) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * part + e;
            if constexpr (!(ABL & 8)) Hf[r >> 3][r & 7] = (half_t)(v[r] * gelu_tab_f(gt[r], phi));
            else Hf[r >> 3][r & 7] = (half_t)(v[r] + gt[r]);
        }
    };
    const std::integral_constant<int, 0> B0{};
    const std::integral_constant<int, 1> B1{};
    // first GEMM, one stage = ten k-steps of BOTH the value and the gate rows of the chunk, fragments interleaved (V_s, G_s): two
    // accumulators alternate, so no MFMA waits for the one issued just before it (20 back-to-back MFMAs on one accumulator with
    // VALU fillers in between would pay the dependent-issue cliff of MI355X_MICROARCH.md at every filler)
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
            if (pv) geglu2(*pv, *pg, half * NG + grp);
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
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chunk = [&](int c, auto cur_, auto prev_) {
        constexpr int cur = decltype(cur_)::value, prev = decltype(prev_)::value;
        // biases of chunk c: value rows at b1[32 c ..], gate rows at b1[inner + 32 c ..]
        init_acc(Hv[cur], sb1 + c * CH);
        init_acc(Hg[cur], sb1 + p.inner + c * CH);
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
    for (int part = 0; part < 8; ++part) geglu2(Hv[1], Hg[1], part);
    {
        const char* s2 = stage_begin();
        gemm2(s2);
    }

    // ---- outputs: fp32 stream and / or the fp16 operand (+ lo plane) of the next GEMM ----
    half_t* o16 = reinterpret_cast<half_t*>(p.out16);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 32 * b + 8 * q + 4 * g;
            const float v[4] = {X[b][4 * q], X[b][4 * q + 1], X[b][4 * q + 2], X[b][4 * q + 3]};
            if constexpr (ABL & 32) { if (v[0] + v[1] + v[2] + v[3] != 12345.678f) continue; }
            if (p.out32) *reinterpret_cast<f32x4*>(p.out32 + row * p.ldo32 + col) = f32x4{v[0], v[1], v[2], v[3]};
            if (o16) {
                const half4v h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4v*>(o16 + row * p.ldo16 + col) = h;
                if (p.out16_lo) store_lo4(p.out16_lo, p.out_lo_fmt, row * p.ldo16 + col, v, h);
            }
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
    if (p.ldx % 4 || p.ldx < p.C || (p.out32 && (p.ldo32 % 4 || p.ldo32 < p.C)) || (p.out16 && (p.ldo16 % 4 || p.ldo16 < p.C))) return PNC_EALIGN;
    if ((((uintptr_t)p.x32 | (uintptr_t)p.tape | (uintptr_t)p.ln_gamma | (uintptr_t)p.ln_beta | (uintptr_t)p.b1 | (uintptr_t)p.b2 |
          (uintptr_t)p.out32) & 15) || (((uintptr_t)p.out16 | (uintptr_t)p.out16_lo) & 7))
        return PNC_EALIGN;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = PNC_OK;
    const float* phi = pnc_gemm::phi_table_device(st, &rc);
    if (rc != PNC_OK) return rc;
    static std::atomic<unsigned char> attr_done[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ff_chain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_done[dev & 63].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL(ff_chain_kernel<0>, dim3(p.M / ROWS), dim3(256), LDS_BYTES, st, p, phi);
    return pnc_launch_status();
}
