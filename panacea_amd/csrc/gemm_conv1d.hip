// gemm_conv1d.hip — PNC_A_CONV1D_T instantiations (temporal nn.Conv1d k=3 of ResBlock3D, openaimodel.py:418,469) of the
// GEMM kernel template.
#include "gemm_kernel.h"

namespace pnc_gemm {

int dispatch_conv1d(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    constexpr int AM = PNC_A_CONV1D_T;
    TileChoice tc = choose_tile(p);
    if (tc.tile == T_128x32) epi = E_GENERIC;
    if (p.gn_part) {
        int rc = PNC_EINVAL;
        bool fused = gn_stats_in_epilogue(p, tc);      // gemm_kernel.h
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
