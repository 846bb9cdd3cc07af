// fa_naive.hip -- rung-0 kernel: one wavefront per query row, fp32 arithmetic on fp32 or bf16 tensors, any head dim <= 256.
//
// Role: the counterpart of the reference's simple kernels (flash_tiled,
// /root/reference/src/flashattention.cu:26-136; silly_attn*, flashattention_lightning.cu:25-264): not tuned, kept
// as an on-device cross-check for the MFMA kernels and -- through FA_KERNEL_AUTO since round 6 -- as the kernel of every head dim they are not
// instantiated for (the reference compiles any d that is a multiple of 32 by editing one macro, flashattention.cu:15,164).
// It runs the same online-softmax recurrence (flashattention.cu:265-342) over chunks of 64 keys:
// lane c scores key c0+c, the row max / row sum are 64-lane butterflies, and P.V walks the chunk with each
// lane owning head-dim columns lane, lane+64, ...
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

constexpr int kNaiveWaves = 4;  // query rows per workgroup

__device__ __forceinline__ float wave_max(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

template <class T>
__device__ __forceinline__ float ld(const T* x) { return (float)*x; }

// TIN: element type of Q, K, V (float or __bf16); TOUT: of O.  The arithmetic is fp32 either way.
template <class TIN, class TOUT>
__global__ __launch_bounds__(kNaiveWaves * kWave) void fa_naive_f32_kernel(FwdParams p, int d, int causal)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // one linear grid: slab-major (grid.y stops at 65 535, the ABI admits 2^31 - 1 slabs)
    const int row_blocks = (p.n + kNaiveWaves - 1) / kNaiveWaves;
    const int row = (int)(blockIdx.x % (unsigned)row_blocks) * kNaiveWaves + wave;
    const int slab = (int)(blockIdx.x / (unsigned)row_blocks);
    if (row >= p.n) return;

    const int b = slab / p.heads, h = slab % p.heads;
    const TIN* q = (const TIN*)p.q + b * p.q_batch_stride + h * p.q_head_stride + (int64_t)row * p.q_row_stride;
    const TIN* kbase = (const TIN*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const TIN* vbase = (const TIN*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    TOUT* o = (TOUT*)p.o + b * p.o_batch_stride + h * p.o_head_stride + (int64_t)row * p.o_row_stride;

    float m = -INFINITY, l = 0.0f;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int lim = causal ? row + 1 : p.n;

    for (int c0 = 0; c0 < lim; c0 += kWave) {
        const int c = c0 + lane;
        float s = -INFINITY;
        if (c < lim) {
            const TIN* kr = kbase + (int64_t)c * p.kv_row_stride;
            float a = 0.0f;
            for (int i = 0; i < d; ++i) a = fmaf(ld(q + i), ld(kr + i), a);
            s = a * p.scale;
        }
        const float mnew = fmaxf(m, wave_max(s));
        const float alpha = expf(m - mnew);  // exp(-inf) = 0 on the first chunk
        const float pr = (c < lim) ? expf(s - mnew) : 0.0f;
        l = l * alpha + wave_sum(pr);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] *= alpha;
        const int cnt = min(kWave, lim - c0);
        for (int t = 0; t < cnt; ++t) {
            const float pt = __shfl(pr, t, 64);
            const TIN* vr = vbase + (int64_t)(c0 + t) * p.kv_row_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = lane + 64 * j;
                if (col < d) acc[j] = fmaf(pt, ld(vr + col), acc[j]);
            }
        }
        m = mnew;
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = lane + 64 * j;
        if (col < d) o[col] = (TOUT)(acc[j] * inv);
    }
    if (p.lse != nullptr && lane == 0) p.lse[(int64_t)slab * p.n + row] = m + logf(l);
}

// dtype: the fa_dtype of the call (0 = fp32 tensors, 1 = bf16 in / bf16 out, 2 = bf16 in / fp32 out)
hipError_t launch_naive(const FwdParams& p, int d, int causal, int dtype, hipStream_t stream)
{
    const int64_t blocks = (int64_t)((p.n + kNaiveWaves - 1) / kNaiveWaves) * p.bh;
    if (blocks > 0x7fffffffLL || d < 1 || d > 256) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(kNaiveWaves * kWave);
    if (dtype == 0) hipLaunchKernelGGL((fa_naive_f32_kernel<float, float>), grid, block, 0, stream, p, d, causal);
    else if (dtype == 1) hipLaunchKernelGGL((fa_naive_f32_kernel<__bf16, __bf16>), grid, block, 0, stream, p, d, causal);
    else if (dtype == 2) hipLaunchKernelGGL((fa_naive_f32_kernel<__bf16, float>), grid, block, 0, stream, p, d, causal);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace fa
