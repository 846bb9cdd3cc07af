// fa_fwd_bf16_x4_p16_causal.hip -- the causal fp16-P instantiations of the x4 kernel; see fa_fwd_bf16_x4_p16.hip.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x4_p16_causal(const FwdParams& p, int out_f32, hipStream_t stream)
{
    return launch_x4_p16<true>(p, out_f32, stream);
}

}  // namespace fa
