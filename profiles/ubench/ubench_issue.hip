// ubench_issue.hip -- per-instruction issue cost on one gfx950 SIMD, alone and next to MFMA from a sibling wave.
// Build: hipcc --offload-arch=gfx950 -O2 ubench_issue.hip -o ubench_issue ; run: ./ubench_issue
// Every block is one CU's worth of waves placed explicitly: blockDim = 64 * 4 * W  (W waves per SIMD).
// Each test body is 32 independent instructions (no dependent chain shorter than 32) repeated ITER times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITER = 2000;

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

template <int OP>
__device__ __forceinline__ void body(float (&r)[32], float k0, float k1)
{
    // 32 instructions per call, each on its own register
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
        if (OP == 0) {  // v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0), "v"(k1));
        } else if (OP == 1) {  // v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        } else if (OP == 2) {  // v_max3_f32
            asm volatile("v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %2, %3" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0), "v"(k1));
        } else if (OP == 3) {  // v_add_f32
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0));
        } else if (OP == 4) {  // v_pk_mul_f32 (2 floats per instruction): one instr covers r[i], r[i+1]
            asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&r[i]) : "v"(*(double*)&r[(i + 2) & 31]));
        } else if (OP == 5) {  // v_pk_add_f32
            asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&r[i]) : "v"(*(double*)&r[(i + 2) & 31]));
        } else if (OP == 6) {  // v_cvt_pk_bf16_f32
            asm volatile("v_cvt_pk_bf16_f32 %0, %0, %2\n\tv_cvt_pk_bf16_f32 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0));
        } else if (OP == 7) {  // v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0));
        } else if (OP == 8) {  // v_pk_fma_f32
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&r[i]) : "v"(*(double*)&r[(i + 2) & 31]));
        } else if (OP == 9) {  // v_exp_f16 (packed? no -- scalar f16 exp)
            asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        } else if (OP == 10) { // v_max_f32
            asm volatile("v_max_f32 %0, %0, %2\n\tv_max_f32 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0));
        } else if (OP == 11) { // v_sub_f32 + v_exp pair pattern: fma then exp on same reg (dependent pair)
            asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %0, %0\n\t" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0), "v"(k1));
        } else if (OP == 12) { // v_pk_mul_f16 as a stand-in for packed 16-bit math
            asm volatile("v_pk_mul_f16 %0, %0, %2\n\tv_pk_mul_f16 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(k0));
        } else if (OP == 13) { // v_ldexp_f32
            asm volatile("v_ldexp_f32 %0, %0, %2\n\tv_ldexp_f32 %1, %1, %2" : "+v"(r[i]), "+v"(r[i + 1]) : "v"(1));
        } else if (OP == 14) { // v_permlane32_swap
            asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        }
    }
}

// role 0: all waves run OP.
// role 1: first-dispatched half of the waves on each SIMD run MFMA only (4 independent accumulators), the rest run OP.
// role 2: as 1 with the halves swapped (MFMA in the younger waves).
// role 3: as 1, OP waves raise their priority (s_setprio 3).      role 4: as 1, MFMA waves raise their priority.
// role 5: as 1, but the MFMA wave uses ONE accumulator (dependent chain).
// role 6: every wave runs [1 MFMA + KV x OP] interleaved in one instruction stream (KV = OP template reuse: see kmix).
template <int OP, int ROLE>
__global__ void k(float* out, unsigned long long* cyc, float k0, float k1)
{
    float r[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = k0 * (i + 1) + threadIdx.x * 1e-6f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves of a workgroup go to SIMDs round-robin, so (wave >> 2) & 1 alternates among the waves sharing one SIMD
    const bool first_half = (((wave >> 2) & 1) == 0);
    const bool do_mfma = (ROLE == 1 || ROLE == 3 || ROLE == 4 || ROLE == 5) ? first_half : (ROLE == 2 ? !first_half : false);
    if (ROLE == 3 && !do_mfma) __builtin_amdgcn_s_setprio(3);
    if (ROLE == 4 && do_mfma) __builtin_amdgcn_s_setprio(3);
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(k0 + i); b[i] = (__bf16)(k1 - i); }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma && ROLE == 5) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < 32; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
    } else if (do_mfma) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < ITER; ++it) body<OP>(r, k0, k1);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += r[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int OP, int ROLE>
void run(const char* name, int W, float* out, unsigned long long* cyc)
{
    const int threads = 64 * 4 * W, blocks = 256;
    hipLaunchKernelGGL((k<OP, ROLE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum_op = 0, sum_mf = 0; int n_op = 0, n_mf = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * W; ++w) {
        const bool fh = (((w >> 2) & 1) == 0);
        const bool mf = (ROLE == 1 || ROLE == 3 || ROLE == 4 || ROLE == 5) ? fh : (ROLE == 2 ? !fh : false);
        if (mf) { sum_mf += h[b * 4 * W + w]; ++n_mf; } else { sum_op += h[b * 4 * W + w]; ++n_op; }
    }
    const double per_op = n_op ? sum_op / n_op / (ITER * 32.0) : 0;      // memtime ticks per instruction, per wave
    const double per_mf = n_mf ? sum_mf / n_mf / (ITER * 32.0) : 0;      // ticks per MFMA, per wave
    printf("%-22s role %d W=%d  ticks/instr/wave %.3f  -> per-SIMD %.3f ticks/instr", name, ROLE, W, per_op,
           per_op / (ROLE >= 1 ? W / 2.0 : W));
    if (n_mf) printf("   | mfma ticks/instr/wave %.2f", per_mf);
    printf("\n");
}

#define RUN_ALL(OP, NAME)                          \
    run<OP, 0>(NAME, 1, out, cyc);                 \
    run<OP, 0>(NAME, 2, out, cyc);                 \
    run<OP, 0>(NAME, 4, out, cyc);                 \
    run<OP, 1>(NAME, 2, out, cyc);                 \
    run<OP, 1>(NAME, 4, out, cyc);

#define RUN_ROLES(OP, NAME)                        \
    run<OP, 2>(NAME, 2, out, cyc);                 \
    run<OP, 3>(NAME, 2, out, cyc);                 \
    run<OP, 4>(NAME, 2, out, cyc);                 \
    run<OP, 5>(NAME, 2, out, cyc);                 \
    run<OP, 2>(NAME, 4, out, cyc);                 \
    run<OP, 3>(NAME, 4, out, cyc);                 \
    run<OP, 5>(NAME, 4, out, cyc);

// one instruction stream: [1 MFMA (2 independent accumulators alternate) + KV v_fma_f32] x 32 per iteration
template <int KV>
__global__ void kmix(float* out, unsigned long long* cyc, float k0, float k1)
{
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = k0 * (i + 1) + threadIdx.x * 1e-6f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(k0 + i); b[i] = (__bf16)(k1 - i); }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < KV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[v % 16]) : "v"(k0), "v"(k1));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < KV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(v + 8) % 16]) : "v"(k0), "v"(k1));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}
template <int KV>
void runmix(int W, float* out, unsigned long long* cyc)
{
    const int threads = 64 * 4 * W, blocks = 256;
    hipLaunchKernelGGL((kmix<KV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4 * W);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += x;
    const double per_mfma = sum / h.size() / (ITER * 32.0);
    printf("mix 1 MFMA + %2d v_fma  W=%d  ticks per (MFMA+VALUs) per wave %.2f -> per SIMD %.2f  (mfma pipe util %.0f%%)\n", KV, W,
           per_mfma, per_mfma / W, 100.0 * 32.0 / (per_mfma / W));
}

int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 256 * 64 * 8);
    // calibrate s_memtime tick: run a known-duration kernel
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); run<0, 0>("warmup v_fma_f32", 4, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    {
        std::vector<unsigned long long> h(16); hipMemcpy(h.data(), cyc, 16 * 8, hipMemcpyDeviceToHost);
        printf("calibration: kernel %.3f ms wall (incl launch), wave ticks %llu -> tick rate >= %.1f MHz\n", ms, h[0], h[0] / (ms * 1e3));
    }
    RUN_ALL(0, "v_fma_f32")
    RUN_ALL(1, "v_exp_f32")
    RUN_ALL(2, "v_max3_f32")
    RUN_ALL(3, "v_add_f32")
    RUN_ALL(7, "v_mul_f32")
    RUN_ALL(10, "v_max_f32")
    RUN_ALL(4, "v_pk_mul_f32")
    RUN_ALL(5, "v_pk_add_f32")
    RUN_ALL(8, "v_pk_fma_f32")
    RUN_ALL(6, "v_cvt_pk_bf16_f32")
    RUN_ALL(9, "v_exp_f16")
    RUN_ALL(12, "v_pk_mul_f16")
    RUN_ALL(13, "v_ldexp_f32")
    RUN_ALL(14, "v_permlane32_swap")
    RUN_ALL(11, "fma->exp dependent")
    printf("---- role variants (2: MFMA in younger waves, 3: VALU waves setprio 3, 4: MFMA waves setprio 3, 5: dependent MFMA chain)\n");
    RUN_ROLES(0, "v_fma_f32")
    RUN_ROLES(1, "v_exp_f32")
    printf("---- single stream interleave\n");
    runmix<0>(1, out, cyc); runmix<2>(1, out, cyc); runmix<4>(1, out, cyc); runmix<6>(1, out, cyc); runmix<8>(1, out, cyc);
    runmix<12>(1, out, cyc); runmix<16>(1, out, cyc);
    runmix<0>(2, out, cyc); runmix<4>(2, out, cyc); runmix<6>(2, out, cyc); runmix<8>(2, out, cyc); runmix<12>(2, out, cyc); runmix<16>(2, out, cyc);
    runmix<6>(4, out, cyc); runmix<8>(4, out, cyc); runmix<12>(4, out, cyc); runmix<16>(4, out, cyc);
    return 0;
}
