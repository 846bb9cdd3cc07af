// fa_fwd_bf16_x2_pb2_d64_bf16out.hip -- the NB = 2 instantiations of the one-wave-per-SIMD kernel with P as bf16 hi + bf16 lo at head dim 64
// (fa_bf16_xn_kernel.h, PF = 3), bf16 output.  One translation unit per head dim and output type: they compile in parallel.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x2_pb2_d64_bf16out(const FwdParams& p, int causal, hipStream_t stream)
{
    if (!xn_addressable(p, 64)) return hipErrorInvalidValue;
    return launch_x2_pb2<64, false>(p, causal, stream);
}

}  // namespace fa
