// fa_fwd_bf16_split.hip -- bf16 tensors through the split kernel (fa_split_kernel.h): the accurate bf16 mode, FA_KERNEL_SPLIT with
// FA_DTYPE_BF16 / FA_DTYPE_BF16_OUT_F32.  Same contract as the bf16 kernels (replaces flash_tiled_coarse{,_causal},
// /root/reference/src/flashattention.cu:139-579, for bf16 tensors); K and V are exact in one bf16 term, Q*scale*log2e and P are
// carried as hi + lo.  out_f32 selects the fp32 or the bf16 output.
#include "fa_split_kernel.h"

namespace fa {

hipError_t launch_bf16_split(const FwdParams& p0, int d, int causal, int out_f32, int mode, hipStream_t stream)
{
    FwdParams p = p0;
    p.o_is_bf16 = out_f32 ? 0 : 1;
    if (mode == 0) mode = choose_split(p, d, causal, 2);
    if ((mode == 3 || mode == 4) && !split_addressable(p, d, 2)) mode = 1;
    switch (d) {
        case 32: return split_launch_bf16_d32(p, causal, mode, stream);
        case 64: return split_launch_bf16_d64(p, causal, mode, stream);
        case 128: return split_launch_bf16_d128(p, causal, mode, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
