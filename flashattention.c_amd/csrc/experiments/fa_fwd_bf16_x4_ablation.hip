// fa_fwd_bf16_x4_ablation.hip -- timing-only ablation instantiations of the x4 kernel (fa_bf16_xn_kernel.h): each switches one
// ingredient of the main loop off (or doubles it) to price it; results are garbage by design.  Used by `fa_driver --variant 33..45`
// and quoted in DESIGN.md section 4.
#include "fa_bf16_xn_kernel.h"

namespace fa {

hipError_t launch_bf16_x4_ablation(const FwdParams& p, int mode, hipStream_t stream)
{
    if (mode == 11) return launch_x4_ablation<1>(p, stream);  // no MFMA
    if (mode == 12) return launch_x4_ablation<2>(p, stream);  // no VALU units
    if (mode == 13) return launch_x4_ablation<3>(p, stream);  // neither: LDS reads, waits, barriers, DMA only
    if (mode == 14) return launch_x4_ablation<4>(p, stream);  // no waits for the V^T fragments
    if (mode == 15) return launch_x4_ablation<8>(p, stream);  // no DMA wait + barrier
    if (mode == 16) return launch_x4_ablation<12>(p, stream); // neither wait
    if (mode == 17) return launch_x4_ablation<16>(p, stream); // no LDS fragment reads
    if (mode == 18) return launch_x4_ablation<17>(p, stream); // no LDS fragment reads, no MFMA
    if (mode == 19) return launch_x4_ablation<32>(p, stream); // every LDS fragment read issued twice (results stay valid)
    if (mode == 20) return launch_x4_ablation<64>(p, stream);  // K fragments not re-read
    if (mode == 21) return launch_x4_ablation<128>(p, stream); // slots not pinned by sched_barrier
    if (mode == 22) return launch_x4_ablation<256>(p, stream); // v_exp_f32 replaced by v_mul_f32
    if (mode == 23) return launch_x4_ablation<512>(p, stream); // no LDS-DMA issue in the loop
    if (mode == 24) return launch_x4_ablation<520>(p, stream); // no LDS-DMA issue, no DMA wait + barrier
    // 30 + k: in-kernel cycle stamps around the fast loop (written to the lse buffer, 4 floats per wave) on top of ablation k
    if (mode == 30) return launch_x4_ablation<1024>(p, stream);
    if (mode == 31) return launch_x4_ablation<1024 + 16>(p, stream);    // no LDS fragment reads
    if (mode == 32) return launch_x4_ablation<1024 + 512>(p, stream);   // no LDS-DMA issue
    if (mode == 33) return launch_x4_ablation<1024 + 2>(p, stream);     // no VALU units
    if (mode == 34) return launch_x4_ablation<1024 + 1>(p, stream);     // no MFMA
    if (mode == 35) return launch_x4_ablation<1024 + 64>(p, stream);    // K fragments not re-read
    if (mode == 36) return launch_x4_ablation<1024 + 8>(p, stream);     // no DMA wait + barrier
    if (mode == 37) return launch_x4_ablation<1024 + 4>(p, stream);     // no waits for the V^T fragments
    if (mode == 38) return launch_x4_ablation<1024 + 512 + 16 + 8>(p, stream);   // no DMA, no LDS reads, no barrier: MFMA + VALU only
    return hipErrorInvalidValue;
}

}  // namespace fa
