// ubench_dot2b.hip -- what does a v_dot2c_f32_bf16 cost beside the matrix pipe: per dot, or per switch between MFMA and dot work?
// One wave per SIMD; per iteration: M MFMAs (32x32x16 bf16, independent accumulators), then N dots back to back, then F v_fma_f32.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dot2b ubench_dot2b.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(4))) float f32x4;
// ALT: 0 = dots behind 32x32x16 MFMAs, 1 = the dot-free residual behind them, 2 = dots behind 16x16x32 MFMAs (the row-sum shape: 16 cycles of pipe)
template <int M, int N, int F, int ALT>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, const float* seed, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 1023];
        b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 517) & 1023];
    }
    float r[16];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed[i] * 0.001f + threadIdx.x * 1e-6f;
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = __float_as_uint(r[i]);
    f32x16 acc[2];
    f32x4 small[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = acc[1][i] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if constexpr (ALT == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(small[m & 1]) : "v"(a), "a"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "a"(b));
            }
#pragma unroll
            for (int f = 0; f < F; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(f + 8) % 16]) : "v"(r[(f + 3) % 16]), "v"(r[(f + 5) % 16]));
            if constexpr (ALT == 0 || ALT == 2) {
#pragma unroll
                for (int n = 0; n < N; ++n) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[n % 8]) : "s"(0x0000bf80u), "v"(u[n % 8]));
            } else {   // the dot-free residual: unpack hi (shift / and), subtract
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    unsigned t;
                    if (n & 1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t) : "v"(u[n % 8]));
                    else asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(t) : "v"(u[n % 8]));
                    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[n % 8]) : "v"(t));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc[0][i] + acc[1][i];
    s += small[0][0] + small[1][1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += __uint_as_float(u[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M, int N, int F, int ALT>
double run(float* out, unsigned long long* cyc, const float* seed)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<M, N, F, ALT>), dim3(256), dim3(256), 0, 0, out, cyc, seed, 100);
    hipLaunchKernelGGL((k<M, N, F, ALT>), dim3(256), dim3(256), 0, 0, out, cyc, seed, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double st = 0;
    for (auto v : h) st += v;
    return st / 1024.0 / (iters * 8.0);
}
#define ROW(M, N, F) printf("%d MFMA, %2d fma, %2d dots : %7.2f cycles   | dot-free residual (%2d x (unpack, sub)): %7.2f\n", M, F, N, run<M, N, F, 0>(out, cyc, seed), N, run<M, N, F, 1>(out, cyc, seed))
int main()
{
    float *out, *seed; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&seed, 4096);
    std::vector<float> h(1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    printf("per iteration: M MFMAs, then F v_fma_f32, then N dots (or the dot-free form)\n");
    ROW(0, 4, 0); ROW(0, 16, 0);
    ROW(1, 0, 0); ROW(1, 1, 0); ROW(1, 2, 0); ROW(1, 4, 0); ROW(1, 8, 0); ROW(1, 16, 0);
    ROW(1, 0, 4); ROW(1, 1, 4); ROW(1, 4, 4); ROW(1, 8, 4); ROW(1, 16, 4);
    ROW(1, 0, 8); ROW(1, 4, 8); ROW(1, 8, 8);
    ROW(2, 0, 0); ROW(2, 4, 0); ROW(2, 8, 0); ROW(2, 16, 0);
    ROW(4, 0, 0); ROW(4, 4, 0); ROW(4, 16, 0);
    printf("behind 16x16x32 MFMAs (16 cycles of pipe): M MFMAs, F fmas, N dots\n");
#define ROW2(M, N, F) printf("%d MFMA 16x16x32, %2d fma, %2d dots : %7.2f cycles\n", M, F, N, run<M, N, F, 2>(out, cyc, seed))
    ROW2(1, 0, 0); ROW2(1, 1, 0); ROW2(1, 4, 0); ROW2(1, 16, 0); ROW2(1, 0, 4); ROW2(1, 4, 4); ROW2(1, 16, 4); ROW2(1, 4, 8); ROW2(2, 0, 0); ROW2(2, 4, 0);
    return 0;
}
