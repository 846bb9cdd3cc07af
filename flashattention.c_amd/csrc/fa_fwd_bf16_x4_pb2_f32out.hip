// fa_fwd_bf16_x4_pb2_f32out.hip -- the NB = 4, D = 64 instantiation of the one-wave-per-SIMD kernel with P as bf16 hi + bf16 lo
// (fa_bf16_xn_kernel.h, PF = 3), fp32 output: the accurate path of large non-causal grids in one launch -- no V copy, no scratch, no
// launch chain.  Holds the 1e-3 bar of the reference comparison (/root/reference/bench_flashattention.py:36-40,74) with P good to ~2^-17.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x4_pb2_f32out(const FwdParams& p, int causal, hipStream_t stream)
{
    if (!xn_addressable(p, 64)) return hipErrorInvalidValue;
    if (causal) return hipErrorInvalidValue;   // 512-row workgroups go to non-causal grids only (bf16_pb2_uses_x4)
    return launch_x4_pb2<false, true>(p, stream);
}

}  // namespace fa
