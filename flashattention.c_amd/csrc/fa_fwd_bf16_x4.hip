// fa_fwd_bf16_x4.hip -- the product tilings of the NB = 4 form of the one-wave-per-SIMD kernel (fa_bf16_xn_kernel.h): D = 64, 128 query
// rows per wave.  Replaces the hot loop of flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:214-354, :434,480-484).
// Non-causal instantiations here, causal ones in fa_fwd_bf16_x4_causal.hip, ablations in fa_fwd_bf16_x4_ablation.hip.
#include "fa_bf16_xn_kernel.h"

namespace fa {

// D = 64 only.  Product library: mode 2 (barrier every two stages, optimistic mix first), non-causal -- the only form the dispatch
// reaches (causal launches take 256-row tiles).  Ablation library: mode 1 (barrier every stage), 3 (lazily rescaled mix only), the causal
// instantiations and modes 11..40, timing-only ablations (results are garbage).
hipError_t launch_bf16_x4(const FwdParams& p, int causal, int out_f32, int mode, hipStream_t stream)
{
    if (!bf16_pipelined_supported(p, 64)) return hipErrorInvalidValue;
#if FA_ABLATION
    if (mode >= 11 && mode <= 40) return launch_bf16_x4_ablation(p, mode, stream);
    if (causal) return launch_bf16_x4_causal(p, out_f32, mode, stream);
    return launch_x4_modes<false>(p, out_f32, mode, stream);
#else
    if (causal || mode != 2) return hipErrorInvalidValue;   // not a tiling of this library
    return launch_x4<2, true, false>(p, out_f32, stream);
#endif
}

}  // namespace fa
