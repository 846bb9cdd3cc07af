// fa_fwd_bf16_x2_p16_d64.hip -- the fp16-P ("accurate") instantiations of the NB = 2 kernel at head dim 64 (fa_bf16_xn_kernel.h; see
// fa_fwd_bf16_x4_p16.hip).  One translation unit per head dim: they compile in parallel.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x2_p16_d64(const FwdParams& p, int causal, int out_f32, hipStream_t stream)
{
    if (!xn_addressable(p, 64)) return hipErrorInvalidValue;
    return launch_x2_p16<64>(p, causal, out_f32, stream);
}

}  // namespace fa
