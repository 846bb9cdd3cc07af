// fa_fwd_bf16_x4_p16x2.hip -- the two-term fp16-P instantiation of the NB = 4 kernel (fa_bf16_xn_kernel.h, PF = 2), D = 64, non-causal (the
// only grids the dispatch gives 512-row workgroups: one well filled round of them, e.g. BASELINE config 4).  See fa_fwd_bf16_x2_p16x2_d64.hip.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x4_p16x2(const FwdParams& p, int out_f32, hipStream_t stream)
{
    if (!bf16_pipelined_supported(p, 64)) return hipErrorInvalidValue;
    return launch_x4_p16<false, 2>(p, out_f32, stream);
}

}  // namespace fa
