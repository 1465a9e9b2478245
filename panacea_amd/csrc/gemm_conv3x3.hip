// gemm_conv3x3.hip — PNC_A_CONV3X3 instantiations (nn.Conv2d 3x3 call-sites incl. stride-2 / nearest-x2 gathers,
// include/panacea_hip.h §1) of the GEMM kernel template.
#include "gemm_kernel.h"

namespace pnc_gemm {

int dispatch_conv3x3(const PncGemmParams& p, unsigned epi, hipStream_t st) {
    constexpr int AM = PNC_A_CONV3X3;
    if (const int geometry = conv3x3_tile_geometry(p, epi)) return dispatch_conv3x3_tiles(p, epi, geometry, st);
    const TileChoice tc = choose_tile(p);
    if (tc.tile == T_128x32 && epi != E_O16 && epi != E_O32) epi = E_GENERIC;
    switch (epi) {
        case E_O16: return launch_tile<AM, E_O16>(p, st, tc);                       // hint stem (SiLU, fp16 between layers)
        case E_O32: return launch_tile<AM, E_O32>(p, st, tc);                       // ResBlock3D in/out layers
        case E_O32 | E_O16: return launch_tile<AM, E_O32 | E_O16>(p, st, tc);       // Down/Upsample feeding a conv
        case E_R1 | E_O32: return launch_tile<AM, E_R1 | E_O32>(p, st, tc);         // first-stage ResnetBlock conv2 + skip
        case E_R1 | E_O32 | E_O16: return launch_tile<AM, E_R1 | E_O32 | E_O16>(p, st, tc);
        default: return launch_tile<AM, E_GENERIC>(p, st, tc);                      // 4 / 3-channel output heads
    }
}

}  // namespace pnc_gemm

PNC_DEFINE_TU_COLLECT(gemm_conv3x3)
