// ubench_regfile.hip -- does the ISSUE cost of v_mfma_f32_32x32x16_bf16 depend on the register files of its operands?
// One wave per SIMD (the x4 regime), an issue-bound stream [1 MFMA + NF v_fma_f32 + NE v_exp_f32] x 16, MFMA forms:
//   acc_a   D/C in AGPRs, A and B in VGPRs          (the P.V products of the attention loop)
//   acc_v   D/C in VGPRs, A in VGPRs, B in AGPRs    (the K.Q^T products: scores where the VALU can read them, Q fragments parked)
//   acc_vv  D/C in VGPRs, A and B in VGPRs
//   first   D in VGPRs, C = 0 (inline constant), A in VGPRs, B in AGPRs   (first k-step of K.Q^T)
// and v_mfma_f32_16x16x32_bf16 (row sums) with D/C in AGPRs.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_regfile ubench_regfile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int FORM, int NF, int NE>
__global__ __launch_bounds__(256, 1) void kreg(float* out, unsigned long long* cyc, const float* seed, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 1023];
        b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 517) & 1023];
    }
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed[i] * 0.001f + threadIdx.x * 1e-6f;
    f32x16 acc0, acc1;
    f32x4 l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
    const float k0 = 1.0001f, k1 = -0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x16& acc = (j & 1) ? acc1 : acc0;
            if constexpr (FORM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
            else if constexpr (FORM == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
            else if constexpr (FORM == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            else if constexpr (FORM == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
            else if constexpr (FORM == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"((j & 1) ? l1 : l0) : "v"(a), "v"(b));
            else if constexpr (FORM == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "a"(b));
#pragma unroll
            for (int v = 0; v < NF; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(v + j) % 16]) : "v"(k0), "v"(k1));
#pragma unroll
            for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(v + j + 8) % 16]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + l0[0] + l1[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int FORM, int NF, int NE>
void run(const char* name, float* out, unsigned long long* cyc, const float* seed)
{
    const int iters = 4000;
    hipLaunchKernelGGL((kreg<FORM, NF, NE>), dim3(256), dim3(256), 0, 0, out, cyc, seed, 100);
    hipLaunchKernelGGL((kreg<FORM, NF, NE>), dim3(256), dim3(256), 0, 0, out, cyc, seed, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double st = 0;
    for (auto v : h) st += v;
    const double per = st / 1024.0 / (iters * 16.0);
    printf("%-8s + %d fma + %d exp : %6.2f cycles per group   (fillers alone would be %3d -> MFMA issue cost %6.2f)\n", name, NF, NE, per,
           4 * NF + 8 * NE, per - (4 * NF + 8 * NE));
}

int main()
{
    float *out, *seed; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&seed, 4096);
    std::vector<float> h(1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
#define ALL(NF, NE)                                   \
    run<0, NF, NE>("acc_a", out, cyc, seed);          \
    run<1, NF, NE>("acc_v", out, cyc, seed);          \
    run<2, NF, NE>("acc_vv", out, cyc, seed);         \
    run<3, NF, NE>("first", out, cyc, seed);          \
    run<5, NF, NE>("all_a", out, cyc, seed);          \
    run<4, NF, NE>("16x16x32", out, cyc, seed);
    ALL(0, 0)
    ALL(4, 4)
    ALL(6, 2)
    ALL(2, 6)
    ALL(8, 0)
    ALL(0, 6)
    return 0;
}
