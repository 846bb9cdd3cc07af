// ubench_dot2.hip -- can the second (lo) bf16 term of P be made from a bf16 hi term with one v_dot2c_f32_bf16 per element?
//   hi_pk = v_cvt_pk_bf16_f32(p0, p1);   r0 = p0 + hi_pk . (-1, 0);   r1 = p1 + hi_pk . (0, -1);   lo_pk = v_cvt_pk_bf16_f32(r0, r1)
// (1) exactness: r must equal p - float(hi) bit for bit (the difference is representable; the question is whether the dot product
//     unit rounds, flushes or reorders anything on the way) over normal, tiny and huge p;
// (2) issue cost of the instructions alone and beside a v_mfma_f32_32x32x16_bf16, one wave per SIMD (same harness as ubench_valu_mix).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_dot2 ubench_dot2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void kexact(const float* p, float* r_dot2c, float* r_dot2, float* r_ref, unsigned* hi_out, unsigned* lo_out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float p0 = p[2 * i], p1 = p[2 * i + 1];
    unsigned hi;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(p0), "v"(p1));
    float a0 = p0, a1 = p1;
    asm volatile("s_nop 1\n\tv_dot2c_f32_bf16 %0, %2, %3\n\tv_dot2c_f32_bf16 %1, %4, %3" : "+v"(a0), "+v"(a1) : "s"(0x0000bf80u), "v"(hi), "s"(0xbf800000u));
    float b0, b1;
    asm volatile("v_dot2_f32_bf16 %0, %2, %3, %5\n\tv_dot2_f32_bf16 %1, %2, %4, %6" : "=&v"(b0), "=&v"(b1) : "v"(hi), "s"(0x0000bf80u), "s"(0xbf800000u), "v"(p0), "v"(p1));
    unsigned lo;
    asm volatile("s_nop 1\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(a0), "v"(a1));
    r_dot2c[2 * i] = a0, r_dot2c[2 * i + 1] = a1;
    r_dot2[2 * i] = b0, r_dot2[2 * i + 1] = b1;
    r_ref[2 * i] = p0 - __uint_as_float(hi << 16);
    r_ref[2 * i + 1] = p1 - __uint_as_float(hi & 0xffff0000u);
    hi_out[i] = hi, lo_out[i] = lo;
}

template <int KIND, bool MFMA>
__global__ __launch_bounds__(256, 1) void kmix(float* out, unsigned long long* cyc, const float* seed, int iters)
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
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x16& acc = (j & 1) ? acc1 : acc0;
            if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
            if constexpr (KIND == 4) {
                // the whole lo sequence of two register pairs behind their hi packs: 2 x (cvt hi, dot2c, dot2c, cvt lo) = 8 instructions
                const int k = (4 * j) % 16, i = (2 * j) % 8;
                asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\t"
                             "v_dot2c_f32_bf16 %2, %6, %0\n\tv_dot2c_f32_bf16 %3, %7, %0\n\tv_dot2c_f32_bf16 %4, %6, %1\n\tv_dot2c_f32_bf16 %5, %7, %1\n\t"
                             "v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5"
                             : "=&v"(u[i]), "=&v"(u[i + 1]), "+v"(r[k]), "+v"(r[k + 1]), "+v"(r[k + 2]), "+v"(r[k + 3])
                             : "s"(0x0000bf80u), "s"(0xbf800000u));
            } else if constexpr (KIND == 5) {
                // the fp16 form of the same work (ships in p16x2): 2 x (cvt hi, v_fma_mixlo, v_fma_mixhi) = 6 instructions
                const int k = (4 * j) % 16, i = (2 * j) % 8;
                asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\t"
                             "v_fma_mixlo_f16 %6, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %6, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                             "v_fma_mixlo_f16 %7, %1, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %7, %1, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "=&v"(u[i]), "=&v"(u[i + 1]), "+v"(r[k]), "+v"(r[k + 1]), "+v"(r[k + 2]), "+v"(r[k + 3]), "=&v"(u[(i + 4) % 8]), "=&v"(u[(i + 5) % 8]));
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = (4 * j + v) % 8, i2 = (4 * j + v + 4) % 8, k = (4 * j + v) % 16;
                    if constexpr (KIND == 0) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[k]) : "s"(0x0000bf80u), "v"(u[i]));
                    else if constexpr (KIND == 1) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(r[k]) : "v"(u[i]), "s"(0x0000bf80u));
                    else if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(r[(k + 3) % 16]), "v"(r[(k + 7) % 16]));
                    else if constexpr (KIND == 3) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[k]) : "v"(u[i2]), "v"(u[i]));
                    else if constexpr (KIND == 6) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[i2]), "v"(r[k]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += __uint_as_float(u[i]);
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
void run(const char* name, int ninst, float* out, unsigned long long* cyc, const float* seed)
{
    const double alone = run1<KIND, false>(out, cyc, seed), with = run1<KIND, true>(out, cyc, seed);
    printf("%d x %-44s : %6.2f cycles alone (%5.2f each)   %6.2f beside one MFMA (32-cycle pipe)\n", ninst, name, alone, alone / ninst, with);
}

int main()
{
    // ---- exactness ----
    const int n = 1 << 22;
    std::vector<float> hp(n);
    unsigned x = 2463534242u;
    for (int i = 0; i < n; ++i) {
        x ^= x << 13, x ^= x >> 17, x ^= x << 5;
        const float mant = 1.0f + (x & 0x7fffff) / 8388608.0f;
        const int cls = (x >> 23) & 7;
        int e;   // exponents: the optimistic mix's whole window 2^-126 .. 2^100, with weight on the ends
        if (cls == 0) e = -126 + (int)((x >> 26) % 12);        // bottom of the normal range (lo goes subnormal / flushes)
        else if (cls == 1) e = 90 + (int)((x >> 26) % 12);
        else e = -100 + (int)((x >> 26) % 64) * 3;
        hp[i] = ldexpf(mant, e);
        if ((i & 1023) == 0) hp[i] = 0.0f;
        if ((i & 1023) == 1) hp[i] = ldexpf(1.0f, e);          // exact powers of two
        if ((i & 1023) == 2) hp[i] = ldexpf(mant, -130 - (int)((x >> 26) % 15));   // subnormal p
    }
    float *dp, *d0, *d1, *d2;
    unsigned *dh, *dl;
    (void)hipMalloc(&dp, n * 4), (void)hipMalloc(&d0, n * 4), (void)hipMalloc(&d1, n * 4), (void)hipMalloc(&d2, n * 4);
    (void)hipMalloc(&dh, n * 2), (void)hipMalloc(&dl, n * 2);
    (void)hipMemcpy(dp, hp.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kexact, dim3(n / 2 / 256), dim3(256), 0, 0, dp, d0, d1, d2, dh, dl, n);
    (void)hipDeviceSynchronize();
    std::vector<float> h0(n), h1(n), h2(n);
    std::vector<unsigned> hh(n / 2), hl(n / 2);
    (void)hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost), (void)hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hh.data(), dh, n * 2, hipMemcpyDeviceToHost), (void)hipMemcpy(hl.data(), dl, n * 2, hipMemcpyDeviceToHost);
    long bad_c = 0, bad_3 = 0, flushed_c = 0, flushed_3 = 0;
    double worst_rel = 0;   // |p - hi - lo| / p over normal p
    for (int i = 0; i < n; ++i) {
        const unsigned pk_h = hh[i / 2], pk_l = hl[i / 2];
        const unsigned hb = (i & 1) ? (pk_h & 0xffff0000u) : (pk_h << 16), lb = (i & 1) ? (pk_l & 0xffff0000u) : (pk_l << 16);
        float hf, lf;
        memcpy(&hf, &hb, 4), memcpy(&lf, &lb, 4);
        const double ref = (double)hp[i] - (double)hf;
        if (memcmp(&h0[i], &h2[i], 4) != 0) {
            if (h0[i] == 0.0f && fabs(ref) < 1.2e-38) ++flushed_c; else { if (bad_c < 5) printf("dot2c mismatch: p=%a hi=%a got %a want %a\n", hp[i], hf, h0[i], h2[i]); ++bad_c; }
        }
        if (memcmp(&h1[i], &h2[i], 4) != 0) {
            if (h1[i] == 0.0f && fabs(ref) < 1.2e-38) ++flushed_3; else { if (bad_3 < 5) printf("dot2 mismatch: p=%a hi=%a got %a want %a\n", hp[i], hf, h1[i], h2[i]); ++bad_3; }
        }
        if (hp[i] > 1e-30f && hp[i] < 1e30f) {
            const double rel = fabs((double)hp[i] - (double)hf - (double)lf) / (double)hp[i];
            if (rel > worst_rel) worst_rel = rel;
        }
    }
    printf("exactness over %d values: v_dot2c_f32_bf16 %ld wrong, %ld flushed-to-zero residuals below 2^-126;  v_dot2_f32_bf16 %ld wrong, %ld flushed\n", n, bad_c,
           flushed_c, bad_3, flushed_3);
    printf("hi + lo against p (1e-30 < p < 1e30): worst relative error %.3e = 2^%.2f\n", worst_rel, log2(worst_rel));

    // ---- issue cost ----
    float *out, *seed; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8); (void)hipMalloc(&seed, 4096);
    std::vector<float> h(1024);
    x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
    (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    run<2>("v_fma_f32 (yardstick)", 4, out, cyc, seed);
    run<0>("v_dot2c_f32_bf16 d, s(-1,0), v", 4, out, cyc, seed);
    run<3>("v_dot2c_f32_bf16 d, v, v", 4, out, cyc, seed);
    run<1>("v_dot2_f32_bf16 d, v, s, d (VOP3P)", 4, out, cyc, seed);
    run<6>("v_fma_mixlo_f16 d, v, -1.0, v", 4, out, cyc, seed);
    run<4>("bf16 pair x2: cvt_pk hi, 2 dot2c, cvt_pk lo", 8, out, cyc, seed);
    run<5>("fp16 pair x2: cvt_pk hi, mixlo, mixhi", 6, out, cyc, seed);
    return 0;
}
