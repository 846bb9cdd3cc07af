// ubench_valu_mix.hip -- issue cost of the VALU instructions of the fp32 three-product kernel's P split, alone and beside an MFMA.
// One wave per SIMD, a stream of [optional v_mfma_f32_32x32x16_bf16 (scores in VGPRs, B in AGPRs) + 4 copies of one instruction kind] x 16.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu_mix ubench_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND, bool MFMA>
__global__ __launch_bounds__(256, 1) void kmix(float* out, unsigned long long* cyc, const float* seed, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)seed[(threadIdx.x * 8 + i) & 1023];
        b[i] = (__bf16)seed[(threadIdx.x * 8 + i + 517) & 1023];
    }
    float r[16];
    f32x2 p[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = seed[i] * 0.001f + threadIdx.x * 1e-6f;
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{r[2 * i], r[2 * i + 1]}, u[i] = __float_as_uint(r[i]);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
    const float k0 = 1.0001f, k1 = -0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x16& acc = (j & 1) ? acc1 : acc0;
            if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int i = (4 * j + v) % 8, i2 = (4 * j + v + 4) % 8, k = (4 * j + v) % 16;
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(k0), "v"(k1));
                else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k]));
                else if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[i2]));
                else if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(p[i2]));
                else if constexpr (KIND == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 5) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(u[i]) : "v"(u[i2]));
                else if constexpr (KIND == 6) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u[i]) : "v"(u[i2]));
                else if constexpr (KIND == 7) asm volatile("s_nop 0");
                else if constexpr (KIND == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 9) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[k]) : "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 12) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]), "s"(0x07060302u));
                else if constexpr (KIND == 13) asm volatile("v_pack_b32_f16 %0, %1, %2 op_sel:[1,1,0]" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 14) asm volatile("v_exp_f16 %0, %0" : "+v"(r[k]));
                else if constexpr (KIND == 15) asm volatile("v_exp_legacy_f32 %0, %0" : "+v"(r[k]));
                else if constexpr (KIND == 16) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 17) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 18) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[i2]));
                else if constexpr (KIND == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[i2]));
                else if constexpr (KIND == 20) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(r[(k + 5) % 16]), "v"(r[(k + 9) % 16]));
                else if constexpr (KIND == 21) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[i2]));
                else if constexpr (KIND == 22) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(k0));
                else if constexpr (KIND == 23) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[i2]));
                else if constexpr (KIND == 24) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(u[i]) : "v"(r[k]));
                else if constexpr (KIND == 25) asm volatile("v_exp_f32 %0, %0 mul:2" : "+v"(r[k]));
                else if constexpr (KIND == 26) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(u[i]) : "v"(u[i2]));
                else if constexpr (KIND == 27) asm volatile("v_alignbit_b32 %0, %1, %2, 16" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 28) asm volatile("v_bfi_b32 %0, %3, %1, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]), "s"(0xffff0000u));
                else if constexpr (KIND == 29) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]), "s"(0xffff0000u));
                else if constexpr (KIND == 30) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 31) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(u[i]) : "v"(r[k]), "v"(r[(k + 5) % 16]));
                else if constexpr (KIND == 32) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(r[k]) : "v"(r[(k + 3) % 16]), "v"(r[(k + 5) % 16]));
                // the softmax element (fma, then exp of its result) in three orders: adjacent pairs, two by two, four by four
                else if constexpr (KIND == 40) asm volatile("v_fma_f32 %1, %0, %2, %3\n\tv_exp_f32 %0, %1" : "+v"(r[k]), "=&v"(u[i]) : "v"(k0), "v"(k1));
                else if constexpr (KIND == 41) { if (v == 0) asm volatile("v_fma_f32 %4, %0, %8, %9\n\tv_fma_f32 %5, %1, %8, %9\n\tv_fma_f32 %6, %2, %8, %9\n\tv_fma_f32 %7, %3, %8, %9\n\t"
                                                                            "v_exp_f32 %0, %4\n\tv_exp_f32 %1, %5\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %7"
                                                                            : "+v"(r[k]), "+v"(r[(k + 1) % 16]), "+v"(r[(k + 2) % 16]), "+v"(r[(k + 3) % 16]), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(k0), "v"(k1)); }
                else if constexpr (KIND == 42) { if ((v & 1) == 0) asm volatile("v_fma_f32 %2, %0, %4, %5\n\tv_fma_f32 %3, %1, %4, %5\n\tv_exp_f32 %0, %2\n\tv_exp_f32 %1, %3"
                                                                            : "+v"(r[k]), "+v"(r[(k + 1) % 16]), "=&v"(u[0]), "=&v"(u[1]) : "v"(k0), "v"(k1)); }
                else if constexpr (KIND == 43) { if (v == 0) asm volatile("v_fma_f32 %4, %0, %8, %9\n\tv_fma_f32 %5, %1, %8, %9\n\tv_exp_f32 %0, %4\n\tv_fma_f32 %6, %2, %8, %9\n\tv_exp_f32 %1, %5\n\tv_fma_f32 %7, %3, %8, %9\n\t"
                                                                            "v_exp_f32 %2, %6\n\tv_exp_f32 %3, %7"
                                                                            : "+v"(r[k]), "+v"(r[(k + 1) % 16]), "+v"(r[(k + 2) % 16]), "+v"(r[(k + 3) % 16]), "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(k0), "v"(k1)); }
                // dependent pairs: the second instruction reads what the first wrote
                else if constexpr (KIND == 10) { if (v & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(r[(k + 15) % 16])); else asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[k]) : "v"(r[(k + 5) % 16])); }
                else if constexpr (KIND == 11) { if (v & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[(k + 7) % 16]) : "v"(r[(k + 15) % 16])); else asm volatile("v_exp_f32 %0, %0" : "+v"(r[k])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + __uint_as_float(u[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, bool MFMA>
double run1(float* out, unsigned long long* cyc, const float* seed)
{
    const int iters = 4000;
    hipLaunchKernelGGL((kmix<KIND, MFMA>), dim3(256), dim3(256), 0, 0, out, cyc, seed, 100);
    hipLaunchKernelGGL((kmix<KIND, MFMA>), dim3(256), dim3(256), 0, 0, out, cyc, seed, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double st = 0;
    for (auto v : h) st += v;
    return st / 1024.0 / (iters * 16.0);
}
template <int KIND>
void run(const char* name, float* out, unsigned long long* cyc, const float* seed)
{
    const double alone = run1<KIND, false>(out, cyc, seed), with = run1<KIND, true>(out, cyc, seed);
    printf("4 x %-34s : %6.2f cycles alone (%5.2f each)   %6.2f beside one MFMA (32-cycle pipe)\n", name, alone, alone / 4, with);
}

int main()
{
    float *out, *seed; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&seed, 4096);
    std::vector<float> h(1024);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>("v_fma_f32", out, cyc, seed);
    run<8>("v_add_f32", out, cyc, seed);
    run<9>("v_sub_f32", out, cyc, seed);
    run<1>("v_exp_f32", out, cyc, seed);
    run<2>("v_pk_add_f32", out, cyc, seed);
    run<3>("v_pk_add_f32 neg", out, cyc, seed);
    run<4>("v_cvt_pk_bf16_f32", out, cyc, seed);
    run<5>("v_and_b32 literal", out, cyc, seed);
    run<6>("v_lshlrev_b32", out, cyc, seed);
    run<7>("s_nop 0", out, cyc, seed);
    run<12>("v_perm_b32", out, cyc, seed);
    run<13>("v_pack_b32_f16 op_sel:[1,1,0]", out, cyc, seed);
    run<14>("v_exp_f16", out, cyc, seed);
    run<15>("v_exp_legacy_f32", out, cyc, seed);
    run<25>("v_exp_f32 mul:2", out, cyc, seed);
    run<16>("v_cvt_pkrtz_f16_f32", out, cyc, seed);
    run<17>("v_cvt_pk_f16_f32", out, cyc, seed);
    run<18>("v_pk_mul_f32", out, cyc, seed);
    run<19>("v_pk_fma_f32", out, cyc, seed);
    run<21>("v_pk_mul_f16", out, cyc, seed);
    run<20>("v_max3_f32", out, cyc, seed);
    run<22>("v_mul_f32", out, cyc, seed);
    run<40>("(fma, exp of it) adjacent      [8 instr]", out, cyc, seed);
    run<42>("(fma, fma, exp, exp)           [8 instr]", out, cyc, seed);
    run<41>("(4 fma, 4 exp)                 [8 instr]", out, cyc, seed);
    run<43>("(fma fma exp fma exp fma exp exp)", out, cyc, seed);
    run<26>("v_mov_b32_sdwa WORD_1 -> WORD_0 keep", out, cyc, seed);
    run<31>("v_or_b32_sdwa src0 WORD_1", out, cyc, seed);
    run<32>("v_add_f32_sdwa (dword sels)", out, cyc, seed);
    run<27>("v_alignbit_b32", out, cyc, seed);
    run<28>("v_bfi_b32", out, cyc, seed);
    run<29>("v_and_or_b32", out, cyc, seed);
    run<30>("v_lshl_or_b32", out, cyc, seed);
    run<23>("v_mov_b32", out, cyc, seed);
    run<24>("v_accvgpr_write_b32", out, cyc, seed);
    run<10>("(v_sub_f32 -> v_cvt_pk dependent) / 2", out, cyc, seed);
    run<11>("(v_exp_f32 -> v_add_f32 dependent) / 2", out, cyc, seed);
    return 0;
}
