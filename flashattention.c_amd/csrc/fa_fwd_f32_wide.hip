// fa_fwd_f32_wide.hip -- the exact fp32 kernel (fa_fwd_f32_kernel.h) at the other head dims the reference can be compiled for: it is
// generic over d % 32 == 0 by editing one macro (/root/reference/src/flashattention.cu:15, num_tiles = d / BK at :164).  FA_KERNEL_AUTO and
// FA_KERNEL_MFMA route fp32 tensors of head dim 96, 160, 192, 224 and 256 here (the split-operand and bf16 families exist at 32, 64, 128
// only; every other head dim up to 256 runs on the rung-0 kernel).  One 128-row workgroup per CU at d >= 192 (two K/V stages of 32 keys
// take 96 - 128 KB of LDS; Q fragments and output accumulators share the wave's 512 registers): 0.79 - 0.85 of the fp32 MFMA peak at B=2 H=8 N=8192.
#include "fa_fwd_f32_kernel.h"

namespace fa {

// io: 0 = fp32 tensors; 1 = bf16 tensors, bf16 output; 2 = bf16 tensors, fp32 output (also the partials of a key-split launch)
hipError_t launch_fwd_f32_wide(const FwdParams& p, int d, int causal, int variant, int io, hipStream_t stream)
{
    if (io != 0) return launch_fwd_f32_wide_bf16(p, d, causal, variant, io == 1 ? 1 : 0, stream);
    switch (d) {
        case 96: return launch_cfg_f32<96, 4, 2>(p, causal, variant, stream);
        case 160: return launch_cfg_f32<160, 4, 1>(p, causal, variant, stream);
        case 192: return launch_cfg_f32<192, 4, 1>(p, causal, variant, stream);
        case 224: return launch_cfg_f32<224, 4, 1>(p, causal, variant, stream);
        case 256: return launch_cfg_f32<256, 4, 1>(p, causal, variant, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
