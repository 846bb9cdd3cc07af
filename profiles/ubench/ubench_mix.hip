// ubench_mix.hip -- one instruction stream per wave: [1 MFMA 32x32x16 + softmax-like VALU mix (+ LDS read)] repeated.
// Mirrors a slot of the attention main loop: NF v_fma, NE v_exp, NC v_cvt_pk_bf16, NX v_max3, optional ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int ITER = 1000;

template <int NF, int NE, int NC, int NX, int LDS>
__global__ void kmix(float* out, unsigned long long* cyc, float k0, float k1)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = k0 * (i + 1) + threadIdx.x * 1e-6f;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(k0 + i); b[i] = (__bf16)(k1 - i); }
    __syncthreads();
    f32x4 ld = {0, 0, 0, 0};
    // three fragment registers in rotation: the LDS read of slot j+2 is issued in slot j (as the attention kernel does)
    bf16x8 fr0 = a, fr1 = a, fr2 = a;
    const unsigned lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (LDS) {
                // issue the read for slot j+2, then wait until the fragment of slot j has landed (2 reads may stay in flight)
                if (j % 3 == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr2) : "v"(lds_addr), "i"((j * 1024) & 8191));
                if (j % 3 == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr0) : "v"(lds_addr), "i"((j * 1024) & 8191));
                if (j % 3 == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr1) : "v"(lds_addr), "i"((j * 1024) & 8191));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 fa = (j % 3 == 0) ? fr0 : (j % 3 == 1) ? fr1 : fr2;
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, b, acc0, 0, 0, 0);
            } else {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NF; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(v + j) % 16]) : "v"(k0), "v"(k1));
#pragma unroll
            for (int v = 0; v < NE; ++v) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(v + j + 4) % 16]));
#pragma unroll
            for (int v = 0; v < NC; ++v) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[(v + j + 8) % 16]) : "v"(k0));
#pragma unroll
            for (int v = 0; v < NX; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[(v + j + 12) % 16]) : "v"(k0), "v"(k1));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = ld[0] + ld[1] + ld[2] + ld[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int NF, int NE, int NC, int NX, int LDS>
void run(int W, float* out, unsigned long long* cyc)
{
    const int threads = 64 * 4 * W, blocks = 256;
    hipLaunchKernelGGL((kmix<NF, NE, NC, NX, LDS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4 * W);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += x;
    const double per = sum / h.size() / (ITER * 16.0);
    printf("1 MFMA + %d fma %d exp %d cvt %d max3 %s  W=%d : %.1f cyc/slot/wave -> %.1f per SIMD (MFMA pipe %.0f%%)\n", NF, NE, NC, NX,
           LDS ? "+ds_read_b128" : "             ", W, per, per / W, 3200.0 / (per / W));
}

int main()
{
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4 * 4); (void)hipMalloc(&cyc, 256 * 64 * 8);
    for (int W : {1, 2, 4}) {
        run<0, 0, 0, 0, 0>(W, out, cyc);
        run<0, 0, 0, 0, 1>(W, out, cyc);
        run<2, 2, 1, 1, 0>(W, out, cyc);
        run<2, 2, 1, 1, 1>(W, out, cyc);
        run<2, 1, 1, 1, 1>(W, out, cyc);
        run<2, 0, 1, 1, 1>(W, out, cyc);
        run<4, 0, 1, 1, 1>(W, out, cyc);
        run<1, 1, 1, 0, 1>(W, out, cyc);
        run<3, 3, 2, 1, 1>(W, out, cyc);
    }
    return 0;
}
