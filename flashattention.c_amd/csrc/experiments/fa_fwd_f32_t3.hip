// fa_fwd_f32_t3.hip -- instantiation and launcher of fa_fwd_f32_t3_kernel (fa_f32_t3_kernel.h): fp32 tensors, head dim 64, long non-causal
// rows, three bf16 products per contraction on pre-split K / V.  An EXPERIMENT of round 2, built into the ablation
// library only: correct (tests run it when that library is present), but no faster than fa_split_kernel.h yet -- its VALU stream is not
// hidden behind the matrix pipe (c3: matrix instructions alone 0.42 ms, + LDS / DMA 0.50, everything 0.66; the split kernel: 0.66).
#include "fa_f32_t3_kernel.h"

namespace fa {

// what the kernel is written for; everything else stays with the split kernel
bool f32_t3_supported(const FwdParams& p, int d, int causal)
{
    return d == 64 && !causal && p.heads == 1 && p.n % 64 == 0 && p.q_row_stride == 64 && p.kv_row_stride == 64 && p.q_batch_stride == (int64_t)p.n * 64 &&
           p.kv_batch_stride == (int64_t)p.n * 64 && ((int64_t)(p.n - 1) * 64 + 64) * 2 < (int64_t)0xffffffffLL;
}

// abl: 0 = the kernel; timing-only ablations (garbage results): 1 no MFMA, 2 no VALU units, 4 no LDS fragment reads, 8 no LDS-DMA in the
// loop, 12 = 4 + 8, 14 = 2 + 4 + 8 (the matrix instructions alone)
hipError_t launch_f32_t3(const FwdParams& p0, int abl, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = 256;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    if (abl == 1) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 1>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else if (abl == 2) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 2>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else if (abl == 4) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 4>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else if (abl == 8) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 8>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else if (abl == 14) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 14>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else if (abl == 12) hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2, 12>), dim3((unsigned)total), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((fa_fwd_f32_t3_kernel<2>), dim3((unsigned)total), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace fa
