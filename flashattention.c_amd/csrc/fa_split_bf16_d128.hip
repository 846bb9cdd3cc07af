// fa_split_bf16_d128.hip -- the split-kernel instantiations for bf16 tensors at head dim 128 (fa_split_kernel.h)
#include "fa_split_kernel.h"

namespace fa {

hipError_t split_launch_bf16_d128(const FwdParams& p, int causal, int mode, hipStream_t stream)
{
    return launch_split_modes<128, true>(p, causal, mode, stream);
}

}  // namespace fa
