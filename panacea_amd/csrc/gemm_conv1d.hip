// gemm_conv1d.hip — PNC_A_CONV1D_T instantiations (temporal nn.Conv1d k=3 of ResBlock3D, openaimodel.py:418,469) of the
// GEMM kernel template.
#include "gemm_kernel.h"

namespace pnc_gemm {

int dispatch_conv1d(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    constexpr int AM = PNC_A_CONV1D_T;
    TileChoice tc = choose_tile(p);
    if (tc.tile == T_128x32) epi = E_GENERIC;
    switch (epi) {
        case E_R1 | E_RB | E_O32: return launch_tile<AM, E_R1 | E_RB | E_O32>(p, st, tc);             // h + conv1d + emb row
        case E_R1 | E_R2 | E_O32: return launch_tile<AM, E_R1 | E_R2 | E_O32>(p, st, tc);             // g + conv1d + skip
        case E_R1 | E_R2 | E_O32 | E_O16: return launch_tile<AM, E_R1 | E_R2 | E_O32 | E_O16>(p, st, tc);
        default: return launch_tile<AM, E_GENERIC>(p, st, tc);
    }
}

}  // namespace pnc_gemm
