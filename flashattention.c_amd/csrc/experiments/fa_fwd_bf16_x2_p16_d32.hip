// fa_fwd_bf16_x2_p16_d32.hip -- the fp16-P ("accurate") instantiations of the NB = 2 kernel at head dim 32 (fa_bf16_xn_kernel.h; see
// fa_fwd_bf16_x4_p16.hip).  One translation unit per head dim: they compile in parallel.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x2_p16_d32(const FwdParams& p, int causal, int out_f32, hipStream_t stream)
{
    if (!xn_addressable(p, 32)) return hipErrorInvalidValue;
    return launch_x2_p16<32>(p, causal, out_f32, stream);
}

}  // namespace fa
