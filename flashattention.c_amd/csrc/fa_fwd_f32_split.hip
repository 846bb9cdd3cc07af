// fa_fwd_f32_split.hip -- fp32 tensors through the split kernel (fa_split_kernel.h): the FA_KERNEL_AUTO path of FA_DTYPE_F32.
// Replaces flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:139-579) for fp32 tensors.
#include "fa_split_kernel.h"

namespace fa {

const char* f32_split_kernel_name() { return "fa_fwd_f32_split_kernel"; }

hipError_t launch_f32_split(const FwdParams& p, int d, int causal, int mode, hipStream_t stream)
{
    if (mode == 0) mode = choose_split(p, d, causal, 4);
    if ((mode == 3 || mode == 4) && !split_addressable(p, d, 4)) mode = 1;
    switch (d) {
        case 32: return split_launch_f32_d32(p, causal, mode, stream);
        case 64: return split_launch_f32_d64(p, causal, mode, stream);
        case 128: return split_launch_f32_d128(p, causal, mode, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
