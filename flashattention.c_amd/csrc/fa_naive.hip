// fa_naive.hip -- rung-0 kernel: one wavefront per query row, fp32, any head dim <= 256.
//
// Role: the counterpart of the reference's simple kernels (flash_tiled,
// /root/reference/src/flashattention.cu:26-136; silly_attn*, flashattention_lightning.cu:25-264): not tuned, kept
// as an on-device cross-check for the MFMA kernels and for head dims they are not instantiated for.
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
    const float* q = (const float*)p.q + b * p.q_batch_stride + h * p.q_head_stride + (int64_t)row * p.q_row_stride;
    const float* kbase = (const float*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const float* vbase = (const float*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    float* o = (float*)p.o + b * p.o_batch_stride + h * p.o_head_stride + (int64_t)row * p.o_row_stride;

    float m = -INFINITY, l = 0.0f;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int lim = causal ? row + 1 : p.n;

    for (int c0 = 0; c0 < lim; c0 += kWave) {
        const int c = c0 + lane;
        float s = -INFINITY;
        if (c < lim) {
            const float* kr = kbase + (int64_t)c * p.kv_row_stride;
            float a = 0.0f;
            for (int i = 0; i < d; ++i) a = fmaf(q[i], kr[i], a);
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
            const float* vr = vbase + (int64_t)(c0 + t) * p.kv_row_stride;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = lane + 64 * j;
                if (col < d) acc[j] = fmaf(pt, vr[col], acc[j]);
            }
        }
        m = mnew;
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = lane + 64 * j;
        if (col < d) o[col] = acc[j] * inv;
    }
    if (p.lse != nullptr && lane == 0) p.lse[(int64_t)slab * p.n + row] = m + logf(l);
}

hipError_t launch_naive_f32(const FwdParams& p, int d, int causal, hipStream_t stream)
{
    const int64_t blocks = (int64_t)((p.n + kNaiveWaves - 1) / kNaiveWaves) * p.bh;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fa_naive_f32_kernel, dim3((unsigned)blocks), dim3(kNaiveWaves * kWave), 0, stream, p, d, causal);
    return hipGetLastError();
}

}  // namespace fa
