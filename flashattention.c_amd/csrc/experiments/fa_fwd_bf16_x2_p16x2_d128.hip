// fa_fwd_bf16_x2_p16x2_d128.hip -- the two-term fp16-P instantiations of the NB = 2 kernel at head dim 128 (fa_bf16_xn_kernel.h, PF = 2):
// bf16 Q, K; P as fp16 hi + fp16 lo and V in fp16 for the second contraction -- the bf16 path that holds the 1e-3 bar of the reference
// comparison (bench_flashattention.py:36-40,74) with two orders of magnitude to spare.  One translation unit per head dim: they
// compile in parallel.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x2_p16x2_d128(const FwdParams& p, int causal, int out_f32, hipStream_t stream)
{
    if (!xn_addressable(p, 128)) return hipErrorInvalidValue;
    return launch_x2_p16<128, 2>(p, causal, out_f32, stream);
}

}  // namespace fa
