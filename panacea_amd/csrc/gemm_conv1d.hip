// gemm_conv1d.hip — PNC_A_CONV1D_T instantiations (temporal nn.Conv1d k=3 of ResBlock3D, openaimodel.py:418,469) of the
// GEMM kernel template.
#include "gemm_kernel.h"

namespace pnc_gemm {

// PncGemmParams.gn_part out of the epilogue: the 256x320 tile's waves (64 rows x 160 columns) own whole groups of N/32 = 10, 20
// or 40 channels, their 64 rows are pixels of one frame, every row and column of a tile exists
static bool gn_stats_in_epilogue(const PncGemmParams& p, const TileChoice& tc) {
    const int cpg = p.N / 32;
    return tc.tile == T_256x320 && (p.N % 320) == 0 && (160 % cpg) == 0 && (cpg % 2) == 0 && (p.Npix % 64) == 0 && (p.M % 64) == 0 &&
           pnc_get_option(PNC_OPT_GEMM_GN_STATS) != 0;
}

int dispatch_conv1d(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    constexpr int AM = PNC_A_CONV1D_T;
    TileChoice tc = choose_tile(p);
    if (tc.tile == T_128x32) epi = E_GENERIC;
    if (p.gn_part) {
        int rc = PNC_EINVAL;
        bool fused = gn_stats_in_epilogue(p, tc);
        if (fused) {
            switch (epi) {
                case E_R1 | E_RB | E_O32: rc = launch<AM, 256, 320, 4, 2, 2, false, E_R1 | E_RB | E_O32 | E_GS>(p, st); break;
                case E_R1 | E_R2 | E_O32: rc = launch<AM, 256, 320, 4, 2, 2, false, E_R1 | E_R2 | E_O32 | E_GS>(p, st); break;
                case E_R1 | E_R2 | E_O32 | E_O16: rc = launch<AM, 256, 320, 4, 2, 2, false, E_R1 | E_R2 | E_O32 | E_O16 | E_GS>(p, st); break;
                default: fused = false;
            }
        }
        if (fused) return rc;
        PncGemmParams q = p;
        q.gn_part = nullptr;
        rc = dispatch_conv1d(q, epi, st);
        if (rc != PNC_OK) return rc;
        return pnc_groupnorm_stats(p.out32, p.ldc32, p.M / p.Npix, p.Npix, p.N, 64, p.gn_part, st);
    }
    switch (epi) {
        case E_R1 | E_RB | E_O32: return launch_tile<AM, E_R1 | E_RB | E_O32>(p, st, tc);             // h + conv1d + emb row
        case E_R1 | E_R2 | E_O32: return launch_tile<AM, E_R1 | E_R2 | E_O32>(p, st, tc);             // g + conv1d + skip
        case E_R1 | E_R2 | E_O32 | E_O16: return launch_tile<AM, E_R1 | E_R2 | E_O32 | E_O16>(p, st, tc);
        default: return launch_tile<AM, E_GENERIC>(p, st, tc);
    }
}

}  // namespace pnc_gemm

PNC_DEFINE_TU_COLLECT(gemm_conv1d)
