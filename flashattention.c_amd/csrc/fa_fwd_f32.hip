// fa_fwd_f32.hip -- fused flash-attention forward in exact fp32 (the reference's dtype) for gfx950.
//
// Replaces flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:139-579) for fp32 tensors.  Both
// contractions run on v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit a k-ordered fmaf chain, at the
// fp32 vector peak (157 TF) while leaving the VALU free for the softmax.  Same skeleton as fa_fwd_bf16.hip:
//
//   workgroup   NWAVES waves x 32 query rows; K/V tiles of 32 keys, LDS-DMA double buffered, one barrier per tile.
//   K image     row-major [key][D] fp32, 16-byte slots XOR-swizzled per row -> conflict-free ds_read_b128 of
//               4 consecutive head-dim values per lane (used by 4 consecutive MFMAs).
//   V image     row-major [key][D] fp32, read column-wise with ds_read_b32 (32 consecutive floats per half-wave).
//   S^T = K Q^T swapped product: lane (q = lane&31, hi) holds scores of keys 4*hi + (r&3) + 8*(r>>2), r = 0..15.
//   O^T += V^T P^T   MFMA #r of a tile contracts over exactly the key pair {r-th key of hi=0, r-th key of hi=1}, so the
//               fp32 P values are fed to the matrix core straight from the registers they were exponentiated in.
//
// The head-dim contraction order of S is d = 8g + 4*hi + e (g-major), i.e. a permutation of the reference's
// d = 0..63 loop (flashattention.cu:236-252); results agree to fp32 rounding, not bitwise.
#include "fa_common.h"
#include "fa_f32_exact.h"
#include "fa_kernels.h"

namespace fa {

template <int D, int NWAVES, bool CAUSAL, int MINWAVES>
__global__ __launch_bounds__(NWAVES* kWave, MINWAVES) void fa_fwd_f32_kernel(FwdParams p)
{
    using C = F32Cfg<D, NWAVES>;
    constexpr int BM = NWAVES * 32;

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];

    if (flag_says_skip(p)) return;   // conditional launch of a chain (the ablation library's chains; FA_KERNEL_AUTO falls back inside the split kernel)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = causal_tile(p, qt);
    const int n = p.n;
    const int q0 = qt * BM + wave * 32;

    const int b = slab / p.heads, h = slab % p.heads;
    const float* qg = (const float*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const float* kg = (const float*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const float* vg = (const float*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab = b * p.o_batch_stride + h * p.o_head_stride;

    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    f32_exact_rows<D, NWAVES, CAUSAL>(p, smem, qg, kg, vg, o_slab, slab, q0, 0, n, kv_end, wave, lane);
}

// MINWAVES_C: the occupancy hint of the causal instantiation (the mask code needs a few registers more: at 4 waves per SIMD, i.e.
// 128 registers, the D = 64 causal kernel spilled 56 bytes per lane)
template <int D, int NWAVES, int MINWAVES, int MINWAVES_C = MINWAVES>
static hipError_t launch_cfg_f32(const FwdParams& p0, int causal, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 32;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    // Three to six workgroups share a CU here.  Dealing a slab's tiles alternately from the heavy and the light end (causal_tile:
    // even rounds of an XCD's workgroups heavy, odd rounds light) evens out what the co-resident workgroups of a CU add up to.
    // Measured, causal, ms plain -> alternating: 16 x 4096 d = 64 0.538 -> 0.328, 8 x 8192 0.787 -> 0.617, 12 x 8192 1.321 -> 1.073,
    // 16 x 8192 d = 32 0.848 -> 0.630, 16 x 8192 d = 64 1.338 -> 1.308, 128 x 1024 0.248 -> 0.236, d = 128 2.45 -> 2.47.
    p.alt_order = causal ? 1 : 0;
    if (causal)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, true, MINWAVES_C>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, false, MINWAVES>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_fwd_f32(const FwdParams& p, int d, int causal, int variant, hipStream_t stream)
{
    (void)variant;
    switch (d) {
        case 32: return launch_cfg_f32<32, 4, 4>(p, causal, stream);
        case 64: return launch_cfg_f32<64, 4, 4, 3>(p, causal, stream);
        case 128: return launch_cfg_f32<128, 4, 2>(p, causal, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
