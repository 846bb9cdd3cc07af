// fa_fwd_bf16_x2.hip -- the NB = 2 instantiations of the one-wave-per-SIMD kernel (fa_bf16_xn_kernel.h): 64 query rows per wave,
// 256-row workgroups, one workgroup per CU; D = 32, 64, 128.  Replaces the hot loop of flash_tiled_coarse{,_causal}
// (/root/reference/src/flashattention.cu:214-354, :434,480-484) for d = 128, d = 32 and the causal / small d = 64 grids.
#include "fa_bf16_xn_kernel.h"

namespace fa {

// d in {32, 64, 128}.  mode 0 = the product configuration (optimistic mix first with the lazily rescaled redo behind it, barrier every 2
// stages) -- the only one in the product library.  Ablation library: 1 = barrier every stage, 3 = lazily rescaled mix only (the same code
// the redo runs), 12 = timing-only ablation without the VALU units (D = 128), 40 = cycle stamps.
hipError_t launch_bf16_x2(const FwdParams& p, int d, int causal, int out_f32, int mode, hipStream_t stream)
{
    if (d != 32 && d != 64 && d != 128) return hipErrorInvalidValue;
    if (!xn_addressable(p, d)) return hipErrorInvalidValue;
#if FA_ABLATION
    if (mode == 40 && d == 64) {   // cycle stamps (fa_driver_ablation --mode prof4 --variant 70): 8 floats per wave into the lse buffer
        FwdParams q;
        dim3 grid, block;
        if (!xn_grid<2>(p, q, grid, block)) return hipErrorInvalidValue;
        if (!causal) (void)xn_launch_order<64, 2>(q, grid, 0, true);
        if (causal) hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<64, 4, true, false, 2, 1024>), grid, block, 0, stream, q);
        else hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<64, 4, false, false, 2, 1024>), grid, block, 0, stream, q);
        return hipGetLastError();
    }
#endif
#if FA_ABLATION
    if (mode == 40 && d == 32 && !causal) {   // ... at d = 32 (round 6: where a short-row launch spends its time)
        FwdParams q;
        dim3 grid, block;
        if (!xn_grid<2>(p, q, grid, block)) return hipErrorInvalidValue;
        (void)xn_launch_order<32, 2>(q, grid, 0, true);
        hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<32, 4, false, false, 2, 1024>), grid, block, 0, stream, q);
        return hipGetLastError();
    }
#endif
#if FA_ABLATION
    if (mode == 41 && d == 64 && causal) {   // timeline (fa_driver_ablation --mode timeline --variant 71): the PRODUCT's launch order, stamped
        FwdParams q;
        dim3 grid, block;
        if (!xn_grid<2>(p, q, grid, block)) return hipErrorInvalidValue;
        const unsigned solo = xn_launch_order<64, 2>(q, grid, causal, true);
        hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<64, 4, true, false, 2, 1024 | 2048>), grid, block, solo, stream, q);
        return hipGetLastError();
    }
#endif
#if FA_ABLATION
    if (mode == 8 && !causal && !out_f32 && d == 64) {   // round 6: EIGHT waves (two per SIMD) sharing the K / V tiles of a 512-row workgroup
        constexpr int BM = 8 * 32 * 2;
        FwdParams q = p;
        q.q_tiles = (p.n + BM - 1) / BM;
        const dim3 grid((unsigned)(q.bh * q.q_tiles)), block(8 * kWave);
        hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<64, 8, false, false, 2, 0, true>), grid, block, 0, stream, q);
        return hipGetLastError();
    }
#endif
#if FA_ABLATION
    if (d == 32 && mode == 1) return launch_x2<32, 1>(p, causal, out_f32, stream);
    if (d == 32 && mode == 3) return launch_x2<32, 2, false>(p, causal, out_f32, stream);
    if (d == 64 && mode == 1) return launch_x2<64, 1>(p, causal, out_f32, stream);
    if (d == 64 && mode == 3) return launch_x2<64, 2, false>(p, causal, out_f32, stream);
    if (d == 128 && mode == 1) return launch_x2<128, 1>(p, causal, out_f32, stream);
    if (d == 128 && mode == 3) return launch_x2<128, 2, false>(p, causal, out_f32, stream);
#endif
    if (mode != 0 && mode != 12) return hipErrorInvalidValue;   // not a tiling of this library
    if (d == 32) return launch_x2<32, 2>(p, causal, out_f32, stream);
    if (d == 64) return launch_x2<64, 2>(p, causal, out_f32, stream);
#if FA_ABLATION
    if (mode == 12) {
        FwdParams q;
        dim3 grid, block;
        if (!xn_grid<2>(p, q, grid, block)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<128, 4, false, false, 2, 2>), grid, block, 0, stream, q);
        return hipGetLastError();
    }
#else
    if (mode == 12) return hipErrorInvalidValue;
#endif
    return launch_x2<128, 2>(p, causal, out_f32, stream);
}

}  // namespace fa
