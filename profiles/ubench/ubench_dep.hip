// ubench_dep.hip -- which register coupling between matrix and vector instructions costs the MFMA/VALU overlap?
// One wave per SIMD; per slot one v_mfma_f32_32x32x16_bf16 and six softmax-like VALU instructions (2 fma, 2 exp, 2 cvt_pk).
//   DST   0: MFMA accumulates into VGPRs            1: into AGPRs
//   READS 0: VALU works on private registers        1: VALU reads (old) results of the VGPR-destination MFMAs
//   FEED  0: MFMA B operand is a constant register  1: B operand is the cvt_pk output of the previous slot
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dep ubench_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int DST, int READS, int FEED>
__global__ __launch_bounds__(256, 1) void kdep(float* out, unsigned long long* cyc, const float* seed, int iters)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 1023];
        b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 517) & 1023];
    }
    f32x16 sv0, sv1, sv2, oa0, oa1, priv;
    for (int i = 0; i < 16; ++i) {
        sv0[i] = seed[i] * 0.01f;
        sv1[i] = seed[i + 16] * 0.01f;
        sv2[i] = seed[i + 48] * 0.01f;
        oa0[i] = 0.0f;
        oa1[i] = 0.0f;
        priv[i] = seed[i + 32] * 0.01f;
    }
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(sv2) : "v"(a), "v"(b));  // an OLD MFMA result
    u32x4 pk = __builtin_bit_cast(u32x4, b);
    const float c = 1.0001f, off = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x16& sd = (j & 1) ? sv1 : sv0;          // destination of this slot's VGPR-form MFMA
            f32x16& sr = (j & 1) ? sv0 : sv1;          // the OTHER tuple: written by the previous slot's MFMA
            const bf16x8 bop = FEED ? __builtin_bit_cast(bf16x8, pk) : b;
            if (DST == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(sd) : "v"(a), "v"(bop));
            } else {
                if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(oa1) : "v"(a), "v"(bop));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(oa0) : "v"(a), "v"(bop));
            }
            // six VALU instructions on two elements
            f32x16& src = READS == 1 ? sr : READS == 2 ? sv2 : priv;
            const int e = 2 * (j % 8);
            float x0 = fmaf(src[e], c, -off), x1 = fmaf(src[e + 1], c, -off);
            x0 = __builtin_amdgcn_exp2f(x0);
            x1 = __builtin_amdgcn_exp2f(x1);
            unsigned w0, w1;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w0) : "v"(x0), "v"(x1));
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w1) : "v"(x1), "v"(x0));
            pk[j & 3] = w0 ^ (w1 & 0x00010001u);
            if (!READS) {
                priv[e] = x0 * 0.999f;  // keep the private chain alive (two more VALU only in this arm would bias it: fold below)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sv0[i] + sv1[i] + sv2[i] + oa0[i] + oa1[i] + priv[i];
    s += (float)pk[0] + (float)pk[1] + (float)pk[2] + (float)pk[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int DST, int READS, int FEED>
void run(float* out, unsigned long long* cyc, const float* seed)
{
    const int iters = 4000;
    hipLaunchKernelGGL((kdep<DST, READS, FEED>), dim3(256), dim3(256), 0, 0, out, cyc, seed, 100);
    hipLaunchKernelGGL((kdep<DST, READS, FEED>), dim3(256), dim3(256), 0, 0, out, cyc, seed, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double st = 0;
    for (auto v : h) st += v;
    printf("dst=%s  valu reads %-14s  B operand %-12s : %6.2f cycles per slot (MFMA alone = 32)\n", DST ? "AGPR" : "VGPR",
           READS == 1 ? "fresh MFMA out" : READS == 2 ? "old MFMA out" : "private regs", FEED ? "from cvt_pk" : "constant", st / 1024 / (iters * 8.0));
}

int main()
{
    float *out, *seed;
    unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 1024 * 8);
    (void)hipMalloc(&seed, 4096);
    std::vector<float> h(1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    run<0, 0, 0>(out, cyc, seed);
    run<0, 1, 0>(out, cyc, seed);
    run<0, 0, 1>(out, cyc, seed);
    run<0, 1, 1>(out, cyc, seed);
    run<1, 0, 0>(out, cyc, seed);
    run<1, 0, 1>(out, cyc, seed);
    run<0, 2, 1>(out, cyc, seed);
    run<1, 2, 1>(out, cyc, seed);
    run<1, 1, 1>(out, cyc, seed);
    return 0;
}
