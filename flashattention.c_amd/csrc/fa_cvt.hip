// fa_cvt.hip -- the bf16 -> fp16 copy of V that precedes the fp16-P kernels (fa_fwd_bf16_x4_p16.hip).
//
// Every bf16 value below 2^16 in magnitude is exact in fp16 down to 2^-14 and loses at most 2^-25 absolutely below that, so the
// copy changes nothing the 1e-3 bar could see; a value of 2^16 or more (or inf / NaN) has no fp16 counterpart, and then the
// chain's flag word is set: the fp16-P kernel skips itself and the split kernel (hi + lo bf16 terms of P) takes the launch.
// HBM-bound: 2 + 2 bytes per element, once per launch (the x4 kernel would otherwise convert each V tile once per q-tile).
#include "fa_kernels.h"

namespace fa {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

__global__ __launch_bounds__(256) void fa_cvt_bf16_to_f16_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t groups,
                                                                  uint32_t* flag, uint32_t serial)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 x = __builtin_nontemporal_load(src + i);
        const bf16x8 b = __builtin_bit_cast(bf16x8, x);
        f16x8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (_Float16)(float)b[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned m = x[e] & 0x7fff7fffu;   // |value| >= 2^16 (bf16 bits >= 0x4780), inf and NaN included
            bad = bad || ((m & 0xffffu) >= 0x4780u) || ((m >> 16) >= 0x4780u);
        }
        dst[i] = __builtin_bit_cast(u32x4, h);
    }
    if (bad && flag != nullptr) __hip_atomic_store(flag, serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// count = number of bf16 elements (a multiple of 8: head dims are); src, dst 16-byte aligned
hipError_t launch_cvt_v_f16(const void* src, void* dst, int64_t count, uint32_t* flag, uint32_t serial, hipStream_t stream)
{
    const int64_t groups = count / 8;
    if (groups < 1 || count % 8 != 0) return hipErrorInvalidValue;
    const int64_t want = (groups + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);   // 8 workgroups per CU, grid-stride
    hipLaunchKernelGGL(fa_cvt_bf16_to_f16_kernel, dim3(grid), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, groups, flag, serial);
    return hipGetLastError();
}

}  // namespace fa
