/*
 * panacea_hip.h — C-ABI of libpanacea_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the Panacea denoising hot path.  The reference
 * (wenyuqing/panacea) ships no native code: every arithmetic call-site on the
 * path is a PyTorch / xformers / cuDNN / cuBLAS call.  Each entry point below
 * replaces one family of those call-sites; the citation after each prototype is
 * the reference call-site (path relative to the reference root) it stands in for.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers (HBM);
 *   - `stream` is a hipStream_t passed as void*; work is only enqueued, never
 *     synchronised; no allocation, no global state -> safe under hipGraph capture;
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or a negative
 *     PNC_E* code for an argument the kernel family does not support;
 *   - activations are channels-last token matrices: row m = (frame f, y, x),
 *     m = (f*H + y)*W + x, columns = channels.  fp16 operands, fp32 residual
 *     stream ("h32") — see DESIGN.md §3.
 */
#ifndef PANACEA_HIP_H
#define PANACEA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNC_OK 0
#define PNC_EINVAL (-1)   /* unsupported shape / argument */
#define PNC_EALIGN (-2)   /* pointer or leading dimension not 16-byte aligned */
#define PNC_EABI (-3)     /* parameter struct compiled against another version of this header */

/* library / build identification: returns "panacea_hip <version> gfx950" */
const char* pnc_version(void);
/* ABI revision of this header (bumped whenever a parameter struct, a prototype or the option list changes): 7
 * (round 4: + pnc_groupnorm_combine, + PNC_OPT_ATTN_DEFER_MAX, PNC_OPT_GEMM_PERSIST is a bit set, - pnc_ff_chain_*;
 *  round 5 (5, 6): + pnc_concat_add_stats, + PNC_OPT_GEMM_STAGGER, + pnc_linear_smallm_segments;
 *  round 6 (7): + PNC_OPT_ATTN_SUM_TRIGGER) */
#define PNC_ABI_VERSION 7
int pnc_abi_version(void);
/* hex SHA-256 of the sources + compile flags the library was built from (panacea_amd/build.py computes the same digest over
 * the checkout): a loader compares the two and refuses a library built from other sources instead of calling it with
 * shifted argument lists */
const char* pnc_build_digest(void);

/* Tuning / test switches (process-global, read with one relaxed atomic load per launch; results never depend on them
 * beyond what each option states).  Returns the previous value, or PNC_EINVAL for an unknown option. */
enum {
    PNC_OPT_GEMM_TAIL_SPLIT = 0,  /* 1 (default): run a sparse last round of output tiles as half / quarter-row workgroups
                                     (bit-identical to 0: rows of a GEMM are independent) */
    PNC_OPT_GEMM_TILE = 1,        /* 0 (default): score-based tile choice; 1 = 128x128, 2 = 256x128, 3 = 256x320,
                                     4 = 256x256 force a geometry where the shape allows it (kernel micro-benchmarks) */
    PNC_OPT_ATTN_VARIANT = 2,     /* 0 (default): by view size; 41 / 81 / 42 / 82 = (waves, query blocks per wave); 42 (large views, default) runs two independent
                                     4-wave workgroups per CU (<= 256 registers); 1 = by view size with 82 (round 3's choice) in the place of 42;
                                     43 = the single-pass few-key kernel (round 6: 64 < kv_valid <= 96 keys shared by all queries of a group —
                                     the text tokens; default there when the grid fills the chip) wherever it applies; any other non-zero
                                     value keeps attn_views_kernel for those launches too (A/B) */
    PNC_OPT_ATTN_DMA = 3,         /* 1 (default): LDS-DMA staging of K / V^T tiles where alignment allows, tile addresses kept as lane constant +
                                     wave-uniform offset; 2 = LDS-DMA with per-tile recomputed addresses (A/B); 0 = register staging.  Same results.
                                     + 4 (round 6, A/B): the few-key launches stay on attn_views_kernel instead of attn_text_kernel */
    PNC_OPT_GEMM_FUSE_LN = 4,     /* bit 0, 1 (default): PncGemmParams.ln_* is reduced in the GEMM epilogue where a workgroup owns whole
                                     rows; 0 = always the LayerNorm kernel after the GEMM (A/B measurements; same result).
                                     + 2 (round 6, A/B): fp32-only epilogues (out32 = acc + bias [+ res1]) of full 256-row tiles go through
                                     the LDS staging, as in round 5, instead of straight from the accumulators.  Bit-identical */
    PNC_OPT_GEMM_GROUP_M = 5,     /* 0 (default): tiles of a GEMM with more than 8 column tiles are walked in groups of 4 row panels
                                     (L2 reuse of W where it exceeds the cache); k > 1 forces groups of k; 1 = plain order.  Results
                                     do not depend on it (same tiles, same arithmetic) */
    PNC_OPT_STENCIL_TILES = 6,    /* 1 (default): stride-1 3x3 convs with Cin % 64 == 0 on grids that fill the chip stage ONE spatial tile of
                                     the input incl. its halo per 64-channel slice and read the nine taps from it (gemm_stencil_tile.hip);
                                     0 = always one gathered A tile per tap; 2 = wherever the shape allows (tests); + 4 (round 6, A/B): the
                                     tile kernel computes its fragment addresses next to the reads (round 5) instead of one MFMA batch
                                     ahead of them.  Bit-identical results either way */
    PNC_OPT_GEMM_PERSIST = 7,     /* bit set, 3 (default).  Bit 0: GEGLU GEMMs of >= 512 full 256x256 tiles run as ONE persistent workgroup
                                     per CU that requests the next output tile's first K tile before its epilogue; bit 1 (round 4): the
                                     same for plain-A launches of >= 512 full 256x320 tiles with a row-major epilogue (residual in place,
                                     fused LayerNorm, fp16 / channel-major outputs, e4m3 lo pass).  Same results either way; 0 = one
                                     tile per workgroup */
    PNC_OPT_ATTN_DEFER_MAX = 8,   /* k (default 8): pnc_attn_views_f16 keeps a query's running softmax maximum — and skips the
                                     rescaling of its accumulators — until some query of the wave exceeds it by more than k in the
                                     exp2 domain (probabilities then reach at most 2^k: fp16-safe up to 15); 0 = rescale whenever a
                                     maximum grows.  Same softmax, other roundings of P (not bit-identical across values) */
    PNC_OPT_GEMM_GN_STATS = 9,    /* 1 (default): PncGemmParams.gn_part comes out of the temporal conv's epilogue where its waves own whole
                                     groups; 0 = always the statistics kernel after the GEMM (same records up to fp32 summation order) */
    PNC_OPT_GEMM_STAGGER = 10,    /* k (default 4 since round 6 — level-0 FF1, K = 320 = 5 tiles, measured 434 -> 420 us staggered; 8 in round 5): the
                                     persistent GEGLU GEMM runs K loops of at least k tiles in the STAGGERED schedule —
                                     four phases per K tile {fragment reads + a third of the next tile's DMA | barrier | MFMAs | barrier},
                                     waves 4-7 one barrier behind waves 0-3, so that on every SIMD one wave multiplies while the other
                                     reads (FF1 at levels 1-2: +5-6 %); 0 = never (round 4's loops); 1 = everywhere the schedule exists —
                                     also the plain-A 8-wave two-stage kernels and the stencil-tile conv kernel, where it measured no
                                     faster (tests, A/B tools).  Same K and MFMA order per accumulator: bit-identical results.
                                     + 256 (round 6, A/B): the persistent GEGLU kernel's products go through its LDS slab (round 3) instead
                                     of straight from the registers through two-byte buffer stores.  Same values */
    PNC_OPT_ATTN_SUM_TRIGGER = 11, /* k (default 12; round 6): after a query block's first K/V tile pnc_attn_views_f16 forms the probabilities
                                     against the running maximum AS IT IS and lets the row sum (needed anyway) tell whether that was safe — a
                                     lane's probabilities are each <= their sum, so sum < 2^k bounds every P below 2^k (fp16-safe up to 14) —
                                     and only a tile whose sum reaches 2^k (or is not finite) computes the row maximum and rescales; 0 = the
                                     row maximum of every tile (round 5).  Same softmax, other roundings of P (as PNC_OPT_ATTN_DEFER_MAX) */
    PNC_OPT_COUNT = 12
};
int pnc_set_option(int option, int value);

/* ------------------------------------------------------------------------- *
 * 1. MFMA GEMM family:  C[M,N] = gatherA[M,K] (fp16) x W[N,K]^T (fp16), fp32 acc
 *
 *   a_mode PNC_A_PLAIN   : A row-major [M][lda]
 *     -> nn.Linear call-sites (sgm/modules/attention.py:94,107,113,220-225,
 *        398-403,509-514,959,980,1040-1059) and 1x1 convs
 *        (sgm/modules/diffusionmodules/openaimodel.py:486, controlmodel.py:81-84)
 *   a_mode PNC_A_CONV3X3 : implicit GEMM over an NHWC fp16 image
 *        [F][Hin][Win][Cin], K = 9*Cin, pad 1, stride 1|2, optional nearest x2
 *        upsample of the input.  K order of W: (ky,kx,ci) when Cin % 64 != 0, else
 *        (ci/64, ky, kx, ci%64) — the nine taps of a 64-channel slice are adjacent
 *        K tiles, so their reads of one pixel neighbourhood hit L1/L2
 *     -> nn.Conv2d 3x3 (openaimodel.py:413,459,125 (Upsample),187 (Downsample),
 *        974 (stem), 1251 (out); controlmodel.py:44-58 (hint stem))
 *   a_mode PNC_A_CONV1D_T: temporal conv1d k=3 pad 1 over frames of one pixel,
 *        rows m = (b*T+t)*Npix + p, K = 3*C ordered (dt,ci), or (ci/64, dt, ci%64) when C % 64 == 0
 *     -> nn.Conv1d (openaimodel.py:418,469) applied on "(b h w) c t"
 *
 *   epilogue (all optional, applied in this order):
 *     v = acc + bias[n] + rowbias[((m / rb_rows) % rb_mod)*N + n]
 *     if geglu: v = value * gelu_erf(gate) on interleaved 32-column blocks
 *               (attention.py:91-98); output has N/2 columns
 *     if act == PNC_ACT_SILU: v = v*sigmoid(v);  PNC_ACT_GELU: v = v*Phi(v)
 *     v += res1[m*ldr1+n] + res2[m*ldr2+n]          (fp32 residual stream)
 *     out32[m*ldc32+n] = v;  out16[m*ldc16+n] = (fp16)v  for n <  n_split
 *     out16t[(m/t_rows)*t_gstride + (n-n_split)*ldt + m%t_rows] = (fp16)v for n >= n_split
 * ------------------------------------------------------------------------- */
enum { PNC_A_PLAIN = 0, PNC_A_CONV3X3 = 1, PNC_A_CONV1D_T = 2 };
/* Storage format of the lo plane of a precise ("split") operand, r = (v - fp16(v)) * 2^11  (PncGemmParams.A_lo):
 *   PNC_LO_F16  : fp16(r), two bytes per element, same leading dimension as the hi plane
 *   PNC_LO_E4M3 : OCP fp8 e4m3 of r clamped to +-448, ONE byte per element, same leading dimension IN ELEMENTS.  |r| <= |v|, so
 *                 no block scale is needed: e4m3 resolves r to 2^-4, i.e. the pair to ~2^-15 of v, and what falls below its
 *                 smallest subnormal (2^-9) is below 2^-20 absolute.  The consumer GEMM runs the lo K loop on the block-scaled
 *                 fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, twice the fp16 rate on half the bytes) with the 2^-11 as the A scale.
 *                 RANGE: r reaches 2^e for |v| in [2^e, 2^(e+1)), so from |v| >= 512 on the clamp at 448 can bite: there the pair
 *                 degrades gracefully (no NaN / Inf) from ~2^-15 of v towards the plain fp16 operand's 2^-11 — for |v| in
 *                 [512, 1024) only the residuals in the top eighth of the rounding interval clamp, from 2048 on most do.
 *                 The denoiser's split operands are normalised activations and fp16 copies of the residual stream; the three
 *                 parity pins (synthetic weights) do not probe the range, and the published checkpoint is not available offline
 *                 to measure its stream.  A caller whose operands reach the hundreds keeps PNC_LO_F16 for them
 *                 (engine.Precision(lo8=False), policy "precise-f16lo").  tests/test_lo8_gpu.py holds values of 700, 1200, 60000. */
enum { PNC_LO_F16 = 0, PNC_LO_E4M3 = 1 };
enum { PNC_ACT_NONE = 0, PNC_ACT_SILU = 1, PNC_ACT_GELU = 2 /* erf GELU: open_clip text-tower MLP */ };

typedef struct PncGemmParams {
    const void* A;          /* fp16 */
    const void* W;          /* fp16 [N][K] row-major (K contiguous) */
    int32_t M, N, K;
    int32_t lda;            /* PNC_A_PLAIN: elements between rows of A */
    int32_t a_mode;
    /* PNC_A_CONV3X3 */
    int32_t Cin, Hin, Win, Hout, Wout, stride, upsample;
    /* PNC_A_CONV1D_T (Cin = channels) */
    int32_t T, Npix;
    /* epilogue */
    const float* bias;      /* [N] or NULL */
    const float* rowbias;   /* [rb_mod][N] or NULL */
    int32_t rb_rows, rb_mod;
    const float* res1; int32_t ldr1;
    const float* res2; int32_t ldr2;
    float*  out32;  int32_t ldc32;
    void*   out16;  int32_t ldc16;
    void*   out16t; int32_t ldt; int32_t t_rows; int64_t t_gstride;
    int32_t n_split;        /* multiple of 128 (or >= N when out16t == NULL) */
    int32_t act;
    int32_t geglu;
    /* split-K workspace, caller-owned (NULL / 0: K is never split).  Small-M shapes (the 4x48 level: M = 3072)
     * fill only 60-120 of the 256 CUs with one K loop per output tile; with a workspace of
     * pnc_gemm_workspace_floats() floats the library runs `s` K-slices per tile and sums the fp32 partials in a
     * fixed order (deterministic) in a second launch that also applies the epilogue. */
    float*  ws;
    int64_t ws_floats;
    /* PNC_A_CONV3X3: 0 = one zero row/column on every side (Conv2d padding=1); 1 = zero padding on the bottom / right
     * only, i.e. F.pad(x, (0,1,0,1)) + Conv2d(padding=0) — the stride-2 Downsample of the first-stage encoder
     * (sgm/modules/diffusionmodules/model.py:108-112) */
    int32_t conv_pad_br;
    /* sizeof(PncGemmParams) as the CALLER compiled it.  pnc_gemm_f16 / pnc_gemm_workspace_floats return PNC_EABI when
     * it differs from the library's, instead of reading past a shorter struct (ABI version 4: 320 bytes). */
    int32_t struct_bytes;
    /* Precise ("split") activation operands.  An fp16 operand v is carried as two fp16 planes,
     *     hi = fp16(v),   lo = fp16((v - hi) * 2^11)        (a 22-bit operand; the 2^11 keeps lo out of the subnormals)
     * A_lo   : lo plane of A (same layout / lda / gather as A), or NULL for a plain fp16 operand.  The kernel runs the
     *          K loop over the lo plane first, scales the accumulators by 2^-11, then runs the hi plane: twice the MFMA
     *          work of the call-site, W is read from L2 twice.
     * out16_lo : lo plane written next to out16 (same ldc16), or NULL.
     * Used by the call-sites whose operand rounding dominates the eps error: the reference
     * computes these contractions in fp32 on CPU / fp16 autocast on GPU (wrappers.py:37-70). */
    const void* A_lo;
    void* out16_lo;
    /* elements between rows of W (0 = K: dense [N][K]).  A strided W lets the K matrix of one attention head, a column
     * block of a [tokens][C] projection buffer, serve as the second operand of S = Q K^T (text tower, first-stage mid block) */
    int32_t ldw;
    /* LayerNorm of the fp32 output rows, fused (attention.py:726-747: x = attn(norm(x)) + x followed by the next norm):
     *   ln_out16[m*ldln + n] = fp16( (out32[m][n] - mean_m) * rstd_m * ln_gamma[n] + ln_beta[n] ),  statistics over the N columns.
     * ln_out16 NULL = off.  Needs out32 and no GEGLU / V^T.  When one workgroup owns whole rows (N <= its tile width: the
     * level-0 width 320) the row statistics are reduced in the epilogue and the normalised fp16 row is written by the GEMM
     * itself (the LayerNorm launch and its 4-byte read of the stream disappear); otherwise the library runs its LayerNorm
     * kernel right after the GEMM on the same stream — same result either way. */
    float ln_eps;
    const float* ln_gamma;
    const float* ln_beta;
    void* ln_out16;
    int32_t ldln;
    /* formats of A_lo / out16_lo (PNC_LO_*).  a_lo_fmt = PNC_LO_E4M3 needs the weight side of the lo pass as well:
     *   W_lo     : e4m3 [N][ldw_lo] bytes, W_lo[n][k] = e4m3(W[n][k] * 2^(127 - w_lo_exp)) in the K order of W
     *   w_lo_exp : E8M0 exponent byte of the tensor, 1..254 (W ~ W_lo * 2^(w_lo_exp - 127)); the packer picks it so that the
     *              tensor maximum lands in [224, 448] — e4m3's 17 binades below that cover every weight that matters
     * and 16-byte chunks of 16 consecutive k: PNC_A_PLAIN lda % 16 == 0 and K % 16 == 0; the conv gathers Cin % 64 == 0. */
    int32_t a_lo_fmt;
    int32_t out_lo_fmt;
    int32_t ldw_lo;             /* bytes between rows of W_lo (0 = K) */
    const void* W_lo;
    int32_t w_lo_exp;
    /* PNC_A_CONV1D_T: 1 = A (and A_lo) hold T + 2 frames per sample, row (b, t, pixel) at ((b (T + 2) + t + 1) Npix + pixel): the
     * frame before the first and after the last of the T frames the rows speak of are present — a frame group's halo frames
     * (engine.FrameShard: the neighbour rank's frame, zeros at the two ends of the clip) — and no tap is padded */
    int32_t t_halo;
    /* PNC_A_CONV3X3 over a BAND of a wider image (engine.ViewShard: a rank's columns of the panorama): 0 = off; otherwise the
     * element offset from A (and from A_lo, in its own elements) to a block [2][frames][Hin][Cin] holding image column -1 (side
     * 0) and column Win (side 1) of every row — the neighbour ranks' edge columns, zeros at the two ends of the panorama.  The
     * taps that leave the band on the left / right read it instead of zero padding; rows above / below the image stay padded.
     * Band, block and everything between them must lie within 2^31 bytes of each frame's origin. */
    int64_t x_halo_off;
    /* PNC_A_CONV1D_T: GroupNorm(32) statistics of the fp32 output, from the GEMM (util.py:276-283: the GroupNorm that follows
     * every temporal conv of a ResBlock3D reads what this launch writes).  NULL = off; otherwise [frames][ceil(Npix / 64)][32][3]
     * floats that receive the {n, mean, M2} records pnc_groupnorm_stats(out32, ldc32, frames, Npix, N, 64, gn_part) would
     * write — frames = M / Npix, channels = N — for pnc_groupnorm_apply(..., n_records = ceil(Npix / 64)).  When a workgroup's
     * waves own whole groups (N % 320 == 0 on the 256x320 tile, Npix % 64 == 0) the records come out of the epilogue, one per
     * (64-row block, group), in a fixed summation order; otherwise the library launches its statistics kernel after the GEMM
     * on the same stream.  Needs out32 (ldc32 % 4 == 0), N % 64 == 0. */
    float* gn_part;
} PncGemmParams;

int pnc_gemm_f16(const PncGemmParams* p, void* stream);
/* 1 when pnc_gemm_f16 would reduce p->ln_* inside the GEMM epilogue for this problem, 0 when it would launch its LayerNorm
 * kernel after the GEMM (a caller that times kernel families separately can then issue pnc_layernorm itself) */
int pnc_gemm_fuses_layernorm(const PncGemmParams* p);
/* floats of workspace with which pnc_gemm_f16 would split K for this problem (0: it would not) */
int64_t pnc_gemm_workspace_floats(const PncGemmParams* p);

/* ------------------------------------------------------------------------- *
 * 2. View-sliced flash attention, head dim 64, fp16 in/out, fp32 softmax/acc.
 *    q rows/kv rows live in token matrices (row m = (group g, y, x)); the query
 *    grid of one group is H x W split along the width into `views` equal slices.
 *    For query view v the key set is the concatenation of kv views
 *    seg[v][0..nseg[v]) of kv group (g / q_per_kv).
 *    K is row-major [rows][ldk]; V is channel-major ("V^T"):
 *        vt[kvg*vt_gstride + (head*64+d)*ldvt + token]
 *    Only the first kv_valid keys of each kv view exist (rest masked).
 *     -> xformers.ops.memory_efficient_attention (attention.py:469 intra-view,
 *        :590 inter-view incl. the view-5 quirk, :363) and
 *        F.scaled_dot_product_attention (attention.py:281) for text keys; with `causal`, the attention of the OpenCLIP
 *        text tower's residual blocks (all heads of all prompts in one launch).
 * ------------------------------------------------------------------------- */
typedef struct PncAttnParams {
    const void* q;  int32_t ldq;    /* fp16, head h at column h*64 */
    const void* k;  int32_t ldk;
    const void* vt; int32_t ldvt; int64_t vt_gstride;
    void* o;        int32_t ldo;
    int32_t groups;                 /* number of q groups (frames) */
    int32_t heads;
    int32_t H, W, views;            /* q grid per group, W % views == 0 */
    int32_t kvH, kvW, kv_views;     /* kv grid per kv group */
    int32_t kv_rows_per_group;      /* rows of K per kv group */
    int32_t q_per_kv;               /* kv group = g / q_per_kv */
    int32_t kv_valid;               /* valid keys per kv view (<= kvH*kvW/kv_views) */
    int32_t nseg[8];                /* per q view */
    int32_t seg[8][2];              /* kv view ids */
    float scale;                    /* softmax scale (d^-0.5) */
    int32_t causal;                 /* 1: query i of a view attends keys j <= i of each kv view only (view-local indices) —
                                       the text tower's causal mask (open_clip build_attention_mask; modules.py:559-632); 0 else */
    /* ABI 5 (round 5): HALO views of a band of views (engine.ViewShard: the cross-view attention of a rank that holds n of the six
     * views attends one view of each neighbour rank as well, attention.py:545-559).  NULL = off.  Otherwise kv view id -1 in seg[][]
     * reads k_halo[0] / vt_halo[0], id kv_views reads k_halo[1] / vt_halo[1]: buffers of EXACTLY the band's geometry (ldk, kvW,
     * kv_rows_per_group, ldvt, vt_gstride) whose view column 0 holds the neighbour's view (the rest is never read).  The band's own
     * K / V^T stay where the QKV GEMM wrote them: no (n + 2)-view copy of keys and values. */
    const void* k_halo[2];
    const void* vt_halo[2];
} PncAttnParams;

int pnc_attn_views_f16(const PncAttnParams* p, void* stream);

/* Temporal self-attention over the T frames of one pixel (head dim 64):
 *   row m = (b*T+t)*Npix + p; q/k/v fp16 with leading dims; out fp16.
 *    -> CrossAttention self-attn on "(b h w) t c" (attention.py:229-291, 1106-1134) */
int pnc_attn_temporal_f16(const void* q, int ldq, const void* k, int ldk,
                          const void* v, int ldv, void* o, int ldo,
                          int B, int T, int Npix, int heads, float scale, void* stream);

/* ------------------------------------------------------------------------- *
 * 3. Normalisations (fp32 stream in, fp16 operand out).  y16_lo (may be NULL) receives the lo plane of a precise
 *    operand, same layout as y16 (see PncGemmParams.A_lo), in the format lo_fmt (PNC_LO_*; pnc_layernorm: fp16 only).
 * ------------------------------------------------------------------------- */
/* Spatial GroupNorm(32,C) over one frame's (H*W, C/32) slab, two launches.
 *   stats: partial[(f*nchunk + chunk)*32 + g] = {count, mean, M2}
 *   apply: y = (x-mean)*rstd*gamma+beta, optional SiLU, -> fp16 [F*Npix][ldy].  n_records: records per frame in `partial`
 *          (0 = ceil(Npix / pix_per_chunk), what pnc_groupnorm_stats with the same chunking wrote; another count when they come
 *          from elsewhere: PncGemmParams.gn_part writes ceil(Npix / 64) per frame) — pix_per_chunk only shapes the apply's own grid
 *    -> nn.GroupNorm (diffusionmodules/util.py:283 eps 1e-5; attention.py:129-132 eps 1e-6) + nn.SiLU */
int pnc_groupnorm_stats(const float* x, int ldx, int F, int Npix, int C,
                        int pix_per_chunk, float* partial, void* stream);
int pnc_groupnorm_apply(const float* x, int ldx, int F, int Npix, int C,
                        int pix_per_chunk, const float* partial,
                        const float* gamma, const float* beta, float eps, int silu,
                        void* y16, int ldy, void* y16_lo, int lo_fmt, int n_records, void* stream);
/* Chan-combine `parts` sets of chunk records, in[((s*F + f)*nchunk + c)*32 + g] = {count, mean, M2} (the all-gathered
 * pnc_groupnorm_stats records of the bands of a view group), per (frame, group), in the fixed order (s, c):
 * out[(f*nchunk + 0)*32 + g] = the combined record, every other slot of the frame {0, 0, 0} — an empty record leaves the
 * combination of pnc_groupnorm_apply unchanged, so that kernel normalises a band with the statistics of the whole panorama.
 *    -> nn.GroupNorm over the (H, 6w) panorama when its views live on several ranks (diffusionmodules/util.py:276-283) */
int pnc_groupnorm_combine(const float* in, int parts, int F, int nchunk, float* out, void* stream);
/* Temporal GroupNorm(32,C)+SiLU: statistics over the (C/32, T) slab of ONE pixel
 *    -> nn.GroupNorm applied on "(b h w) c t" (openaimodel.py:409-419,509-515) */
int pnc_groupnorm_temporal_silu(const float* x, int B, int T, int Npix, int C,
                                const float* gamma, const float* beta, float eps,
                                void* y16, void* y16_lo, int lo_fmt, void* stream);
/* The same split over the ranks of a frame group (engine.FrameShard, round 4), for the T local frames of every pixel:
 *   mode 1: stats[((b*Npix + p)*32 + g)*2 + {0,1}] = {sum, sum of squares} over the (C/32, T) values this rank holds;
 *   mode 2: normalise + SiLU with `stats` = those sums added over the ranks and T_total = frames per sample over all ranks;
 *           t_pad = 1 writes y into the (T + 2)-frame layout PncGemmParams.t_halo reads (frame t at slot t + 1).
 *    -> the same nn.GroupNorm on "(b h w) c t" when the T frames of a pixel live on several ranks */
int pnc_groupnorm_temporal_part(const float* x, int B, int T, int Npix, int C,
                                const float* gamma, const float* beta, float eps,
                                float* stats, int mode, int T_total,
                                void* y16, void* y16_lo, int lo_fmt, int t_pad, void* stream);
/* LayerNorm over C (eps 1e-5) -> nn.LayerNorm (attention.py:699-701) */
int pnc_layernorm(const float* x, int ldx, int M, int C,
                  const float* gamma, const float* beta, float eps,
                  void* y16, int ldy, void* y16_lo, void* stream);

/* ------------------------------------------------------------------------- *
 * 4. Small helpers
 * ------------------------------------------------------------------------- */
/* out[m][n] = sum_k f(a[m][k]) * W[n][k] + bias[n], f = SiLU if silu_in; a fp32,
 * W fp16, M <= 16.  -> time_embed / emb_layers Linear (openaimodel.py:936-943,440-447) */
int pnc_linear_smallm(const float* a, int lda, const void* W, const float* bias,
                      float* out, int ldo, int M, int N, int K, int silu_in, int silu_out,
                      void* stream);
/* The same linear for nseg sites in ONE launch (ABI 6): W [N][K] holds the sites' weight rows back to back, seg_start[0 .. nseg]
 * (HOST array, ascending, seg_start[0] = 0, seg_start[nseg] = N, every entry % 4 == 0, nseg <= PNC_SMALLM_MAX_SEGS) names each
 * site's column range, and the output is one contiguous [Mtot][width_s] block per site, blocks back to back:
 *     out[seg_start[s] * Mtot + (m0 + m) * width_s + (n - seg_start[s])],  m < M <= 16 rows of this call, m0 + M <= Mtot
 * — the [rows][N_site] row-bias operand each site's GEMM epilogue reads.  Per column bit-identical to pnc_linear_smallm.
 * -> the emb_layers Linear of every ResBlock of a network after one time_embed (openaimodel.py:440-447, 1296-1298) */
#define PNC_SMALLM_MAX_SEGS 64
int pnc_linear_smallm_segments(const float* a, int lda, const void* W, const float* bias, float* out, int M,
                               int m0, int Mtot, int N, int K, const int32_t* seg_start, int nseg, int silu_in,
                               int silu_out, void* stream);
/* sinusoidal timestep embedding out[f] = [cos(t*freqs) | sin(t*freqs)], fp32; freqs[dim/2] is
 * tabulated by the caller  (diffusionmodules/util.py:224-248) */
int pnc_timestep_embedding(const int64_t* t, int F, int dim, const float* freqs,
                           float* out, void* stream);
/* NCHW (fp32) -> channels-last fp16 with optional second source (channel concat) and zero padding to Cpad:
 *   out[f][p][c] = c<C1 ? a[f % a_frames][c][p]*a_scale[f] : c<C1+C2 ? b[f][c-C1][p] : 0
 * a_scale NULL = 1: the per-frame c_in of DiscreteDenoiser (denoiser.py:27-28) folded into the conversion; a_frames < F:
 * `a` holds the latent once and both CFG halves read it (torch.cat([x] * 2), guiders.py:36, without the copy)
 *    -> torch.cat in wrappers.py:41 + layout change */
int pnc_nchw_to_tokens_f16(const float* a, int C1, const float* a_scale, int a_frames, const float* b, int C2,
                           int F, int Npix, int Cpad, void* out16, void* out16_lo, void* stream);
/* Exit of one sampler step on the network's channels-last eps, one pass, reference rounding order:
 *   D_h    = eps_h * c_out + x                       DiscreteDenoiser + EpsScaling: c_out = -sigma snapped to the table,
 *                                                    c_skip = 1 (denoiser.py:22-28, denoiser_scaling.py:16-22)
 *   D      = D_u + scale * (D_c - D_u)               VanillaCFG (guiders.py:25-29); cfg = 0: D = D_c, one half only
 *   x_next = x + (sigma_next - sigma) * ((x - D) / sigma)      EulerEDMSampler.sampler_step (sampling.py:96-133)
 * eps_tok [(cfg ? 2 : 1) * T * Npix][ld] fp32, uncond frames first; x, x_next NCHW [T][C][Npix]; c_out, sigma, sigma_next [T]. */
int pnc_cfg_euler_step(const float* eps_tok, int ld, int T, int Npix, int C, int cfg, float scale,
                       const float* x, const float* c_out, const float* sigma, const float* sigma_next, float* x_next,
                       void* stream);
/* channels-last fp32 [F*Npix][ld] -> NCHW fp32 (first C columns) */
int pnc_tokens_to_nchw_f32(const float* x, int ld, int F, int Npix, int C,
                           float* out, void* stream);
/* out32[m][0:C1] = a[m][:], out32[m][C1:C1+C2] = s[m][:] + c[m][:]; optional fp16 copy
 *    -> th.cat([h, hs.pop() + control.pop()], dim=1)  (controlmodel.py:193-195) */
int pnc_concat_add(const float* a, int C1, const float* s, const float* c, int C2,
                   int64_t M, float* out32, void* out16, void* out16_lo, int lo_fmt, void* stream);
/* the same concatenation over F frames of Npix pixels (M = F * Npix rows) that ALSO writes the GroupNorm(32) statistics of its output —
 * records {n, mean, M2} [F][ceil(Npix / pix_per_chunk)][32][3] as pnc_groupnorm_stats(..., pix_per_chunk, ...) would, for
 * pnc_groupnorm_apply(..., n_records = ceil(Npix / pix_per_chunk)): the first GroupNorm of the ResBlock3D that follows the concat
 * (openaimodel.py:1311-1314 -> 499-503) then needs no statistics launch.  (C1 + C2) % 64 == 0.  Deterministic (no float atomics).
 * ABI 5 (round 5). */
int pnc_concat_add_stats(const float* a, int C1, const float* s, const float* c, int C2, int F, int Npix, int pix_per_chunk,
                         float* out32, void* out16, void* out16_lo, int lo_fmt, float* partial, void* stream);
/* y = x + a (fp32, may be in place); optional fp16 copy of y
 *    -> h += guided_hint / h += control.pop()  (controlmodel.py:127,192) */
int pnc_add_f32(const float* x, const float* a, int64_t n, float* y32, void* y16, void* y16_lo, int lo_fmt,
                void* stream);
/* fp32 -> fp16 (+ lo plane in lo_fmt) */
int pnc_cast_f16(const float* x, int64_t n, void* y16, void* y16_lo, int lo_fmt, void* stream);

/* Range monitor (round 6; ABI 7).  The eps contract of the `precise` operand policy is written for a residual stream inside the e4m3
 * lo plane's range (|v| < 512: beyond it the lo plane of a split operand saturates at +-448 and the operand falls back to fp16's 11
 * bits — UNetModel3D.eps_contract; the reference has no counterpart: its fp32 CPU path has no such range and its
 * autocast fp16 path overflows at 65504 without notice, sgm/modules/diffusionmodules/wrappers.py:37-70).  Every kernel that packs an
 * e4m3 lo plane counts the packed quads that clamped; this call ADDS the counts since the previous call to *out (a device word the
 * caller zeroes) on `stream` and resets them — a handful of one-thread launches, once per network evaluation.  0 = the evaluation
 * stayed inside the range its contract is written for. */
int pnc_range_monitor_collect(unsigned int* out, void* stream);

/* ------------------------------------------------------------------------- *
 * 5. First-stage decoder: row softmax of a materialised score
 *    matrix, p[m][:] = softmax(scale * s[m][:]) (fp32 in, fp32 statistics, fp16
 *    out), N <= 16384, N % 4 == 0.  Columns >= n_valid (n_valid <= 0: N) are masked; causal != 0 also masks columns
 *    j > m (row m = query m of ONE sequence: the OpenCLIP text tower's attn_mask, sgm/modules/encoders/modules.py:608).  The single-head d = 512 attention of the VAE
 *    mid block runs as  S = Q K^T (pnc_gemm_f16, fp32 out) -> this -> O = P V
 *    (pnc_gemm_f16 against the channel-major V^T).
 *     -> AttnBlock / MemoryEfficientAttnBlock.attention
 *        (sgm/modules/diffusionmodules/model.py:393-408, 444-472)
 * ------------------------------------------------------------------------- */
int pnc_softmax_rows_f16(const float* s, int64_t lds, int M, int N, float scale, int causal, int n_valid,
                         void* p16, int64_t ldp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PANACEA_HIP_H */
