// fa_fwd_bf16_x4_causal.hip -- the causal instantiations of the x4 kernel (fa_bf16_xn_kernel.h); see fa_fwd_bf16_x4.hip.
// Replaces flash_tiled_coarse_causal (/root/reference/src/flashattention.cu:434,480-484) for large D = 64 grids.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x4_causal(const FwdParams& p, int out_f32, int mode, hipStream_t stream)
{
    return launch_x4_modes<true>(p, out_f32, mode, stream);
}

}  // namespace fa
