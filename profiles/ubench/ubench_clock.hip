// ubench_clock.hip -- what does the matrix pipe deliver in REAL time?  s_memtime (shader clock) against s_memrealtime
// (constant 100 MHz) for MFMA-only and MFMA+VALU streams, zero vs random operands, all 256 CUs busy.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_clock ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV, int NE, int NP = 0>
__global__ void kclk(float* out, unsigned long long* cyc, const float* seed, int iters)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 1023];
        b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 517) & 1023];
    }
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed[i] * 0.001f + threadIdx.x * 1e-6f;
    f32x16 acc0 = {0}, acc1 = {0};
    const float k0 = 1.0001f, k1 = -0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(v + j) % 16]) : "v"(k0), "v"(k1));
#pragma unroll
            for (int v = 0; v < NP; ++v) {
                typedef __attribute__((ext_vector_type(2))) float f2;
                f2 t = {r[(2 * v + j) % 16 & ~1], r[((2 * v + j) % 16 & ~1) + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"((f2){k0, k0}), "v"((f2){k1, k1}));
                r[(2 * v + j) % 16 & ~1] = t[0];
                r[((2 * v + j) % 16 & ~1) + 1] = t[1];
            }
#pragma unroll
            for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(v + j + 8) % 16]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + wave;
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
}

template <int NV, int NE, int NP = 0>
void run(const char* data, int W, float* out, unsigned long long* cyc, const float* seed)
{
    const int threads = 64 * 4 * W, blocks = 256, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((kclk<NV, NE, NP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, seed, 200);   // warm
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((kclk<NV, NE, NP>), dim3(blocks), dim3(threads), 0, 0, out, cyc, seed, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4 * W * 2);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double st = 0, sr = 0;
    for (size_t i = 0; i < h.size(); i += 2) { st += h[i]; sr += h[i + 1]; }
    const double nw = h.size() / 2.0;
    const double ticks = st / nw, real_us = sr / nw / 100.0;        // s_memrealtime = 100 MHz
    const double mfma = iters * 16.0;
    const double tflops = 256.0 * 4 * W * mfma * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-6s 1 MFMA + %d fma + %d pk_fma + %d exp  W=%d : %6.2f ticks/MFMA/SIMD, memtime %7.1f MHz, event %.3f ms (in-kernel %.3f ms) -> %7.1f TF\n", data, NV,
           NP, NE, W, ticks / mfma / W, ticks / real_us, ms, real_us * 1e-3, tflops);
}

int main()
{
    float *out, *seed_r, *seed_z; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4 * 4); (void)hipMalloc(&cyc, 256 * 64 * 16);
    (void)hipMalloc(&seed_r, 4096); (void)hipMalloc(&seed_z, 4096);
    std::vector<float> h(1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed_r, h.data(), 4096, hipMemcpyHostToDevice);
    (void)hipMemset(seed_z, 0, 4096);
    for (int rep = 0; rep < 2; ++rep) {
        for (int W : {1, 2}) {
            run<0, 0>("zero", W, out, cyc, seed_z);
            run<0, 0>("random", W, out, cyc, seed_r);
            run<4, 0>("random", W, out, cyc, seed_r);
            run<4, 2>("random", W, out, cyc, seed_r);
            run<3, 3>("random", W, out, cyc, seed_r);
            run<6, 6>("random", W, out, cyc, seed_r);
            run<4, 4>("random", W, out, cyc, seed_r);
            run<0, 4, 2>("random", W, out, cyc, seed_r);
            run<2, 2>("random", W, out, cyc, seed_r);
            run<0, 2, 1>("random", W, out, cyc, seed_r);
        }
    }
    return 0;
}
