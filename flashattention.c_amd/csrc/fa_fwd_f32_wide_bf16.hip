// fa_fwd_f32_wide_bf16.hip -- the exact fp32 kernel (fa_fwd_f32_kernel.h) on BF16 tensors at head dims 96, 160, 192, 224, 256; bf16 or fp32 output
// (FwdParams::o_is_bf16, a uniform branch around the stores).  The bf16 MFMA families exist at 32, 64, 128; until round 6 bf16 tensors of any other head dim ran
// on the rung-0 kernel -- 246 ms at 16 x 8192 x 96 where the same call on fp32 tensors takes 3.3.  Arithmetic: fp32 (v_mfma_f32_32x32x2_f32), the
// tensors widened on their way into the fp32 LDS images (fa_f32_exact.h).
#include "fa_fwd_f32_kernel.h"

namespace fa {

hipError_t launch_fwd_f32_wide_bf16(const FwdParams& p, int d, int causal, int variant, int out_bf16, hipStream_t stream)
{
    FwdParams q = p;
    q.o_is_bf16 = out_bf16;
    switch (d) {
        case 96: return launch_cfg_f32<96, 4, 2, 2, __bf16>(q, causal, variant, stream);
        case 160: return launch_cfg_f32<160, 4, 1, 1, __bf16>(q, causal, variant, stream);
        case 192: return launch_cfg_f32<192, 4, 1, 1, __bf16>(q, causal, variant, stream);
        case 224: return launch_cfg_f32<224, 4, 1, 1, __bf16>(q, causal, variant, stream);
        case 256: return launch_cfg_f32<256, 4, 1, 1, __bf16>(q, causal, variant, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
