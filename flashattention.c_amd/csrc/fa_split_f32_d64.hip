// fa_split_f32_d64.hip -- the split-kernel instantiations for fp32 tensors at head dim 64 (fa_split_kernel.h)
#include "fa_split_kernel.h"

namespace fa {

hipError_t split_launch_f32_d64(const FwdParams& p, int causal, int mode, hipStream_t stream)
{
    return launch_split_modes<64, false>(p, causal, mode, stream);
}

}  // namespace fa
