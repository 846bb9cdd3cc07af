// fa_fwd_f32.hip -- the exact fp32 kernel (fa_fwd_f32_kernel.h) at the head dims every family is instantiated for: 32, 64, 128.
// Replaces flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:139-579) for fp32 tensors under FA_KERNEL_MFMA.
#include "fa_fwd_f32_kernel.h"

namespace fa {

hipError_t launch_fwd_f32(const FwdParams& p, int d, int causal, int variant, hipStream_t stream, int io)
{
    if (variant < 0 || variant > 2 || (variant == 2 && !causal)) return hipErrorInvalidValue;   // 1 / 2: one tile per workgroup / paired tiles
    if (io != 0) return launch_fwd_f32_wide(p, d, causal, variant, io, stream);                  // bf16 tensors: the wide head dims only
    switch (d) {
        case 32: return launch_cfg_f32<32, 4, 4>(p, causal, variant, stream);
        case 64: return launch_cfg_f32<64, 4, 4, 3>(p, causal, variant, stream);
        case 128: return launch_cfg_f32<128, 4, 2>(p, causal, variant, stream);
        default: return launch_fwd_f32_wide(p, d, causal, variant, 0, stream);
    }
}

}  // namespace fa
