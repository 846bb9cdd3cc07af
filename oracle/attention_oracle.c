/*
 * oracle/attention_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the attention forward that kilianhae/FlashAttention.C computes.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (flashattention.c_amd) never links or calls it.
 *
 * Parity pinning (see oracle/README.md, DESIGN.md section "Oracle"):
 *   - oracle_attention_f64 / oracle_attention_f32 are checked against the reference's own
 *     Python oracle functions (bench_flashattention.py:36-48), executed in the build
 *     container by oracle/make_golden.py, and against the golden vectors that script
 *     committed under tests/golden/.
 *   - oracle_flash_tiled_f32 restates the tiled online-softmax recurrence of the CUDA kernel
 *     (src/flashattention.cu:214-354) and is checked against oracle_attention_f64.
 *   - oracle_attention_packed_f32 restates src/llm.c/attention_forward.cu:53-125 and is
 *     checked against that very function compiled from the reference tree into
 *     oracle/_ref/ (oracle/Makefile target `ref`).
 *
 * Conventions shared with the reference:
 *   tensors are (BH, N, d) row-major fp32; element (b, r, c) lives at b*N*d + r*d + c
 *   (src/flashattention.cu:144,198,224,350); `scale` multiplies S before the max
 *   (src/flashattention.cu:258-263) and the reference passes scale = 1.0 (:593).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* -------------------------------------------------------------------------------------------
 * 1. Direct attention, fp64 accumulate.
 *    O[b,r,:] = sum_c softmax_c(scale * <Q[b,r,:], K[b,c,:]>) * V[b,c,:]
 *    follows bench_flashattention.py:36-40 (unmasked) and :42-48 (causal: keys c > r get
 *    -inf before the softmax).  The reference's masked variant builds the mask with
 *    `tril == 0`, which also masks an exactly-zero score inside the triangle (SURVEY.md
 *    section 4); that quirk is not reproduced -- for continuous random inputs it never fires.
 *    If lse != NULL it receives log(sum_c exp(s_c)) per row (natural log, includes scale);
 *    this is the quantity the reference reserves `O_l` for (src/flashattention.cu:609) but
 *    never fills.
 * ----------------------------------------------------------------------------------------- */
void oracle_attention_f64(const float* q, const float* k, const float* v, double* o, double* lse,
                          int64_t bh, int64_t n, int64_t d, double scale, int causal)
{
#pragma omp parallel
    {
        double* s = (double*)malloc((size_t)n * sizeof(double));
#pragma omp for collapse(2) schedule(static)
        for (int64_t b = 0; b < bh; ++b) {
            for (int64_t r = 0; r < n; ++r) {
                const float* qr = q + (b * n + r) * d;
                const int64_t lim = causal ? (r + 1) : n;
                double m = -INFINITY;
                for (int64_t c = 0; c < lim; ++c) {
                    const float* kc = k + (b * n + c) * d;
                    double acc = 0.0;
                    for (int64_t i = 0; i < d; ++i) acc += (double)qr[i] * (double)kc[i];
                    acc *= scale;
                    s[c] = acc;
                    if (acc > m) m = acc;
                }
                double l = 0.0;
                for (int64_t c = 0; c < lim; ++c) {
                    s[c] = exp(s[c] - m);
                    l += s[c];
                }
                double* orow = o + (b * n + r) * d;
                for (int64_t i = 0; i < d; ++i) orow[i] = 0.0;
                for (int64_t c = 0; c < lim; ++c) {
                    const float* vc = v + (b * n + c) * d;
                    const double p = s[c];
                    for (int64_t i = 0; i < d; ++i) orow[i] += p * (double)vc[i];
                }
                const double inv = 1.0 / l;
                for (int64_t i = 0; i < d; ++i) orow[i] *= inv;
                if (lse) lse[b * n + r] = m + log(l);
            }
        }
        free(s);
    }
}

/* Same contract, fp32 result (rounded once from the fp64 computation). */
void oracle_attention_f32(const float* q, const float* k, const float* v, float* o, float* lse,
                          int64_t bh, int64_t n, int64_t d, float scale, int causal)
{
    const size_t ne = (size_t)bh * (size_t)n * (size_t)d;
    double* od = (double*)malloc(ne * sizeof(double));
    double* ld = lse ? (double*)malloc((size_t)bh * (size_t)n * sizeof(double)) : NULL;
    oracle_attention_f64(q, k, v, od, ld, bh, n, d, (double)scale, causal);
    for (size_t i = 0; i < ne; ++i) o[i] = (float)od[i];
    if (lse) {
        for (size_t i = 0; i < (size_t)bh * (size_t)n; ++i) lse[i] = (float)ld[i];
        free(ld);
    }
    free(od);
}

/* -------------------------------------------------------------------------------------------
 * 2. The reference kernel's own recurrence, in fp32, tile by tile (tile = 32 key rows).
 *    Follows the math contract of flash_tiled_coarse (src/flashattention.cu:214-354):
 *      for each key tile j (:214):    S = scale * Q_i K_j^T                    (:217-263)
 *        m' = max(m, rowmax S)  (:265-274);  a = exp(m - m')
 *        O *= a (skipped for j == 0, :277-285);  l *= a (:288-290)
 *        p = exp(S - m');  l += p;  O += p V_j                                  (:313-342)
 *      out = O / l                                                              (:346-354)
 *    causal variant: tile loop stops at the diagonal tile (:434) and masks col > row to
 *    -inf inside it (:480-484).
 *    Deliberate difference: key rows past n are masked with -inf here, where the CUDA
 *    kernel zero-fills them (:224-231) and is therefore only correct for n % 32 == 0
 *    (SURVEY.md F8).  For n % 32 == 0 the two are the same recurrence.
 * ----------------------------------------------------------------------------------------- */
#define ORACLE_BC 32
void oracle_flash_tiled_f32(const float* q, const float* k, const float* v, float* o,
                            int64_t bh, int64_t n, int64_t d, float scale, int causal)
{
#pragma omp parallel
    {
        float* acc = (float*)malloc((size_t)d * sizeof(float));
#pragma omp for collapse(2) schedule(static)
        for (int64_t b = 0; b < bh; ++b) {
            for (int64_t r = 0; r < n; ++r) {
                const float* qr = q + (b * n + r) * d;
                float m = -INFINITY, l = 0.0f;
                float s[ORACLE_BC];
                for (int64_t i = 0; i < d; ++i) acc[i] = 0.0f;
                const int64_t ntiles = (n + ORACLE_BC - 1) / ORACLE_BC;
                const int64_t jend = causal ? (r / ORACLE_BC + 1) : ntiles;
                for (int64_t j = 0; j < jend; ++j) {
                    float tmax = -INFINITY;
                    for (int c = 0; c < ORACLE_BC; ++c) {
                        const int64_t kc = j * ORACLE_BC + c;
                        float sv;
                        if (kc >= n || (causal && kc > r)) {
                            sv = -INFINITY;
                        } else {
                            const float* kr = k + (b * n + kc) * d;
                            float a = 0.0f;
                            for (int64_t i = 0; i < d; ++i) a += qr[i] * kr[i];
                            sv = a * scale;
                        }
                        s[c] = sv;
                        if (sv > tmax) tmax = sv;
                    }
                    const float mnew = tmax > m ? tmax : m;
                    const float alpha = expf(m - mnew); /* exp(-inf) = 0 on the first tile */
                    if (j > 0)
                        for (int64_t i = 0; i < d; ++i) acc[i] *= alpha;
                    l *= alpha;
                    for (int c = 0; c < ORACLE_BC; ++c) {
                        const int64_t kc = j * ORACLE_BC + c;
                        if (s[c] == -INFINITY) continue;
                        const float p = expf(s[c] - mnew);
                        l += p;
                        const float* vr = v + (b * n + kc) * d;
                        for (int64_t i = 0; i < d; ++i) acc[i] += p * vr[i];
                    }
                    m = mnew;
                }
                float* orow = o + (b * n + r) * d;
                for (int64_t i = 0; i < d; ++i) orow[i] = acc[i] / l;
            }
        }
        free(acc);
    }
}

/* -------------------------------------------------------------------------------------------
 * 3. llm.c-layout causal attention.
 *    inp is (B, T, 3C) with Q | K | V concatenated on the last axis, heads interleaved inside
 *    each C-wide slab; out is (B, T, C); scale = 1/sqrt(hs); causal.
 *    Follows src/llm.c/attention_forward.cu:53-125 pass by pass (running max seeded with
 *    -10000 at :71, expsum guard at :100); preatt/att scratch buffers are not materialised.
 * ----------------------------------------------------------------------------------------- */
void oracle_attention_packed_f32(const float* inp, float* out, int B, int T, int C, int NH)
{
    const int C3 = 3 * C;
    const int hs = C / NH;
    const float scale = 1.0f / sqrtf((float)hs);
#pragma omp parallel
    {
        float* att = (float*)malloc((size_t)T * sizeof(float));
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b) {
            for (int t = 0; t < T; ++t) {
                for (int h = 0; h < NH; ++h) {
                    const float* query = inp + (size_t)b * T * C3 + (size_t)t * C3 + h * hs;
                    float maxval = -10000.0f;
                    for (int t2 = 0; t2 <= t; ++t2) {
                        const float* key = inp + (size_t)b * T * C3 + (size_t)t2 * C3 + h * hs + C;
                        float val = 0.0f;
                        for (int i = 0; i < hs; ++i) val += query[i] * key[i];
                        val *= scale;
                        if (val > maxval) maxval = val;
                        att[t2] = val;
                    }
                    float expsum = 0.0f;
                    for (int t2 = 0; t2 <= t; ++t2) {
                        const float e = expf(att[t2] - maxval);
                        expsum += e;
                        att[t2] = e;
                    }
                    const float inv = expsum == 0.0f ? 0.0f : 1.0f / expsum;
                    float* orow = out + (size_t)b * T * C + (size_t)t * C + h * hs;
                    for (int i = 0; i < hs; ++i) orow[i] = 0.0f;
                    for (int t2 = 0; t2 <= t; ++t2) {
                        const float* value = inp + (size_t)b * T * C3 + (size_t)t2 * C3 + h * hs + 2 * C;
                        const float a = att[t2] * inv;
                        for (int i = 0; i < hs; ++i) orow[i] += a * value[i];
                    }
                }
            }
        }
        free(att);
    }
}

/* Threads the OpenMP build will use (1 for the serial build); reported as cpu_baseline.cores. */
int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
