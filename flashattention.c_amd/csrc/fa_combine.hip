// fa_combine.hip -- the merge step of a key-split launch (FlashDecoding-style): S workgroups per q-tile have left normalised fp32
// partial outputs and their log-sum-exps in the workspace; this kernel weighs them together.  Grids that leave the chip idle only
// (fa_plan.cpp: keysplit_factor); counterpart of nothing in the reference, whose grid simply runs any (BH, N)
// (/root/reference/src/flashattention.cu:592,599).
#include "fa_kernels.h"

namespace fa {


// ---- combine of a key-split launch (fa_launch.cpp: launch_bf16_keysplit) -------------------------------------------------------
// S workgroups per q-tile each saw a share of the keys and left a normalised partial output O_s (fp32, [S][bh][n][d]) and the
// log-sum-exp of its share (natural log, [bh][S][n]; -inf for a causal share that lies entirely above the row).  O = sum_s w_s O_s / sum_s w_s with w_s = exp(lse_s - max_s lse_s);
// lse = max + log(sum w_s).  One thread per four output columns; HBM-bound and small (S * 4 bytes per output element).
// Every load of a thread is issued before the first is used (S <= kMaxSplits, unrolled: the first form of this kernel walked the shares
// behind a data-dependent `continue` -- S memory round trips in a row, 8.9 us for one slab of 8192 rows where this form takes 4).
constexpr int kMaxSplits = 8;
template <bool OUT_F32>
__global__ __launch_bounds__(256) void fa_combine_splits_kernel(FwdParams p, const float* __restrict__ o_part, const float* __restrict__ lse_part,
                                                                 int S, int d)
{
    const int tpr = d / 4;                                       // threads per row
    const int64_t rows = (int64_t)p.bh * p.n;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row_id = t / tpr;
    if (row_id >= rows) return;
    const int c = (int)(t % tpr) * 4;
    const int slab = (int)(row_id / p.n), row = (int)(row_id % p.n);
    float l[kMaxSplits];
    f32x4 x[kMaxSplits];
#pragma unroll
    for (int s = 0; s < kMaxSplits; ++s) {
        if (s < S) {   // uniform
            l[s] = lse_part[((int64_t)slab * S + s) * p.n + row];
            x[s] = *(const f32x4*)(o_part + (((int64_t)s * p.bh + slab) * p.n + row) * d + c);
        } else {
            l[s] = -INFINITY;
            x[s] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    float m = l[0];
#pragma unroll
    for (int s = 1; s < kMaxSplits; ++s) m = fmaxf(m, l[s]);
    float wsum = 0.0f;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < kMaxSplits; ++s) {
        const float w = __expf(l[s] - m);
        // an empty share (causal launches: its keys lie above this row's tile; lse = -inf) never wrote its O: whatever was loaded is dropped
        const bool live = w != 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += live ? w * x[s][e] : 0.0f;
        wsum += w;
    }
    const float inv = 1.0f / wsum;
    const int b = slab / p.heads, h = slab % p.heads;
    const int64_t off = b * p.o_batch_stride + h * p.o_head_stride + (int64_t)row * p.o_row_stride + c;
    if constexpr (OUT_F32) {
        *(f32x4*)((float*)p.o + off) = acc * inv;
    } else {
        bf16x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (__bf16)(acc[e] * inv);
        *(bf16x4*)((__bf16*)p.o + off) = r;
    }
    if (p.lse != nullptr && c == 0) p.lse[(int64_t)slab * p.n + row] = m + __logf(wsum);
}

hipError_t launch_combine_splits(const FwdParams& p, const float* o_part, const float* lse_part, int S, int d, int out_f32, hipStream_t stream)
{
    if (S < 1 || S > kMaxSplits) return hipErrorInvalidValue;
    const int64_t threads = (int64_t)p.bh * p.n * (d / 4);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (out_f32)
        hipLaunchKernelGGL(fa_combine_splits_kernel<true>, dim3(blocks), dim3(256), 0, stream, p, o_part, lse_part, S, d);
    else
        hipLaunchKernelGGL(fa_combine_splits_kernel<false>, dim3(blocks), dim3(256), 0, stream, p, o_part, lse_part, S, d);
    return hipGetLastError();
}

}  // namespace fa
