// gemm_plain.hip — PNC_A_PLAIN instantiations (nn.Linear / 1x1 conv call-sites, include/panacea_hip.h §1) of the GEMM
// kernel template, one per (tile geometry, epilogue variant) the denoising path uses; everything else runs E_GENERIC.
#include "gemm_kernel.h"

namespace pnc_gemm {

// E_LN variants exist for the two geometries whose workgroup can own a whole row of the path's widths
template <unsigned EPI>
static int launch_ln(const PncGemmParams& p, hipStream_t st, TileChoice tc) {
    if (tc.tile == T_256x320) return launch<PNC_A_PLAIN, 256, 320, 4, 2, 2, false, EPI>(p, st);
    return launch<PNC_A_PLAIN, 128, 128, 2, 2, 2, true, EPI>(p, st);
}

int dispatch_plain(const PncGemmParams& p, unsigned epi, hipStream_t st, bool* ln_fused) {
    constexpr int AM = PNC_A_PLAIN;
    const TileChoice tc = choose_tile(p);
    if (epi & E_LN) {
        // fuse only when ONE column tile covers the row (level-0 width 320 on 256x320; <= 128 on 128x128)
        if (ln_whole_rows(p, tc)) *ln_fused = true;
        else epi &= ~E_LN;                      // the caller runs the LayerNorm kernel after this GEMM instead
    }
    if (tc.tile == T_128x32 && epi != E_O16 && epi != E_O32) epi = E_GENERIC;     // narrow-N: two fast variants
    if (tc.tile == T_256x320 && tc.ksplit == 1 && plain_persist_ok(p, epi)) {
        // level-0 launches of the fp32-epilogue families: one persistent workgroup per CU, next tile's first K tile under the epilogue
        switch (epi) {
            case E_O16: return launch_plain_persist<E_O16>(p, st);
            case E_O16 | E_VT: return launch_plain_persist<E_O16 | E_VT>(p, st);
            case E_O32: return launch_plain_persist<E_O32>(p, st);
            case E_R1 | E_O32: return launch_plain_persist<E_R1 | E_O32>(p, st);
            case E_R1 | E_O16: return launch_plain_persist<E_R1 | E_O16>(p, st);
            case E_O32 | E_LN: return launch_plain_persist<E_O32 | E_LN>(p, st);
            case E_RB | E_O32 | E_LN: return launch_plain_persist<E_RB | E_O32 | E_LN>(p, st);
            case E_R1 | E_O32 | E_LN: return launch_plain_persist<E_R1 | E_O32 | E_LN>(p, st);
            default: break;
        }
    }
    if (epi == (E_O16 | E_GELU) && tc.tile != T_256x256 && tc.tile != T_128x128) epi = E_GENERIC;
    switch (epi) {
        case E_O16: return launch_tile<AM, E_O16>(p, st, tc);                       // q (text), qkv (temporal), text K
        case E_O16 | E_VT: return launch_tile<AM, E_O16 | E_VT>(p, st, tc);         // qkv of the view attention, text V^T
        case E_O32: return launch_tile<AM, E_O32>(p, st, tc);                       // proj_in, skip 1x1, zero convs
        case E_O32 | E_O16: return launch_tile<AM, E_O32 | E_O16>(p, st, tc);
        case E_R1 | E_O32: return launch_tile<AM, E_R1 | E_O32>(p, st, tc);         // to_out, ff2, proj_out (+ residual, in place)
        case E_R1 | E_O32 | E_O16: return launch_tile<AM, E_R1 | E_O32 | E_O16>(p, st, tc);
        case E_R1 | E_O16: return launch_tile<AM, E_R1 | E_O16>(p, st, tc);         // ff2 of the last block: fp16 for proj_out
        case E_RB | E_O32: return launch_tile<AM, E_RB | E_O32>(p, st, tc);         // proj_in_temporal + position table
        case E_O32 | E_LN: return launch_ln<E_O32 | E_LN>(p, st, tc);               // proj_in + norm1
        case E_RB | E_O32 | E_LN: return launch_ln<E_RB | E_O32 | E_LN>(p, st, tc); // proj_in_temporal + pos + norm1
        case E_R1 | E_O32 | E_LN: return launch_ln<E_R1 | E_O32 | E_LN>(p, st, tc); // to_out + residual + norm2 / norm3
        case E_GEGLU | E_O16:                                                       // ff1
            if (tc.tile == T_256x256 && geglu_persist_ok(p)) return launch_geglu_persist<256, 256, 4, 2>(p, st);
            return launch_tile<AM, E_GEGLU | E_O16>(p, st, tc);
        case E_O16 | E_GELU:                                                        // text-tower c_fc + GELU (two geometries)
            if (tc.tile == T_256x256) return launch<AM, 256, 256, 4, 2, 2, true, E_O16 | E_GELU>(p, st);
            return launch<AM, 128, 128, 2, 2, 2, true, E_O16 | E_GELU>(p, st);
        default: return launch_tile<AM, E_GENERIC>(p, st, tc);
    }
}

}  // namespace pnc_gemm

PNC_DEFINE_TU_COLLECT(gemm_plain)
