// fa_driver.cpp -- torch-less host driver for the C ABI (counterpart of /root/reference/test.cu:606-646).
//
// The reference driver hard-codes N = 8192, batch = 8, fills Q = K = iota and V = 1, launches the causal kernel
// once and prints a wall-clock time; it performs no check (and no longer compiles at HEAD).
// This driver keeps that workload as `--mode kat` and turns its implied known answer into a check: every softmax
// row sums to 1 and V == 1, so O must be exactly 1 wherever it is written.  `--mode rand` times random data
// (never time constant data: DVFS inflates it) and cross-checks the MFMA kernel against the rung-0 kernel ON THE
// DEVICE; `--mode sweep` does that for every co-compiled tiling variant.  Prints one JSON object per run.
//
//   fa_driver --mode kat  [--bh 8] [--n 8192] [--d 64] [--dtype f32|bf16] [--causal 1]
//   fa_driver --mode rand [--bh 16] [--n 8192] [--d 64] [--dtype bf16] [--causal 0] [--scale 1.0] [--iters 20] [--variant 0]
//   fa_driver --mode sweep ...
//   fa_driver --mode llmc [--B 6] [--T 4096] [--C 768] [--NH 12] [--iters 100]
//       the llm.c dev harness (/root/reference/src/llm.c/attention_forward.cu:1209-1295) at its own size: srand(0)-style U(-1, 1)
//       activations in the packed (B, T, 3C) layout, fa_forward_packed_qkv, every output element checked against the rung-0 kernel
//       ON THE DEVICE at the harness' own 1e-4 (validate_result, :1255-1262), then the mean of `iters` launches between two events on
//       the null stream (benchmark_kernel, src/llm.c/common.h:108-124).
//   fa_driver --mode sharded [--devices N] [--bh 1024] [--n 8192] [--d 64] [--dtype bf16] [--iters 5]
//       BASELINE config 5's arithmetic on whatever devices are visible: the batch*head axis cut into N contiguous shards (the
//       first bh % N shards one slab longer), one fa_forward_sharded call per iteration, per-device milliseconds from events on each
//       device's stream and their maximum (the job time: no collective on the path).  FA_ALLOW_SAME_DEVICE=1 lets N exceed the number
//       of devices (shards then share devices round-robin): the only way to exercise N > 1 on a one-GPU box.
//   --kernel auto|mfma|split|pb2 (ablation driver: also p16|p16x2) and --out_f32 1 choose the kernel family / an fp32 output for bf16 tensors in rand and sweep mode.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "flashattn_amd.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

static uint16_t f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Rng {  // xorshift64* + Box-Muller: deterministic N(0,1) without libc rand()
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
    uint64_t next()
    {
        s ^= s >> 12;
        s ^= s << 25;
        s ^= s >> 27;
        return s * 0x2545F4914F6CDD1Dull;
    }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    float normal()
    {
        const double u1 = uni(), u2 = uni();
        return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    }
};

struct Args {
    std::string mode = "rand", dtype = "bf16", kernel = "";
    long bh = 16, n = 8192;
    int d = 64, causal = 0, iters = 20, warmup = 3, variant = 0, check = 1, out_f32 = 0;
    int devices = 0;                         // sharded mode: number of shards (0 = every visible device)
    int B = 6, T = 4096, C = 768, NH = 12;   // llmc mode: the harness' own size (attention_forward.cu:1217-1220)
    bool iters_given = false;
    float scale = 1.0f;
};

static Args parse(int argc, char** argv)
{
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        auto val = [&]() -> const char* {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", k.c_str());
                exit(2);
            }
            return argv[++i];
        };
        if (k == "--mode") a.mode = val();
        else if (k == "--dtype") a.dtype = val();
        else if (k == "--bh") a.bh = atol(val());
        else if (k == "--n") a.n = atol(val());
        else if (k == "--d") a.d = atoi(val());
        else if (k == "--causal") a.causal = atoi(val());
        else if (k == "--iters") a.iters = atoi(val()), a.iters_given = true;
        else if (k == "--kernel") a.kernel = val();
        else if (k == "--out_f32") a.out_f32 = atoi(val());
        else if (k == "--B") a.B = atoi(val());
        else if (k == "--T") a.T = atoi(val());
        else if (k == "--C") a.C = atoi(val());
        else if (k == "--NH") a.NH = atoi(val());
        else if (k == "--warmup") a.warmup = atoi(val());
        else if (k == "--variant") a.variant = atoi(val());
        else if (k == "--check") a.check = atoi(val());
        else if (k == "--scale") a.scale = (float)atof(val());
        else if (k == "--devices") a.devices = atoi(val());
        else {
            fprintf(stderr, "unknown flag %s\n", k.c_str());
            exit(2);
        }
    }
    return a;
}

struct Buffers {
    void *q = nullptr, *k = nullptr, *v = nullptr, *o = nullptr;        // in the run dtype
    float *qf = nullptr, *kf = nullptr, *vf = nullptr, *of = nullptr;   // fp32 copies for the on-device cross-check
    size_t ne = 0;
};

static void upload(const Args& a, const std::vector<float>& hq, const std::vector<float>& hk, const std::vector<float>& hv,
                   Buffers& b, bool want_f32_copy)
{
    const bool bf = a.dtype == "bf16";
    b.ne = hq.size();
    const size_t esz = bf ? 2 : 4;
    HIP_OK(hipMalloc(&b.q, b.ne * esz));
    HIP_OK(hipMalloc(&b.k, b.ne * esz));
    HIP_OK(hipMalloc(&b.v, b.ne * esz));
    HIP_OK(hipMalloc(&b.o, b.ne * 4));         // large enough for an fp32 output of bf16 tensors (--out_f32)
    HIP_OK(hipMemset(b.o, 0xff, b.ne * 4));    // poison: unwritten output shows up as NaN
    std::vector<float> rq(hq), rk(hk), rv(hv);  // values as the kernel sees them
    if (bf) {
        std::vector<uint16_t> t(b.ne);
        const std::vector<float>* src[3] = {&hq, &hk, &hv};
        std::vector<float>* dst[3] = {&rq, &rk, &rv};
        void* dev[3] = {b.q, b.k, b.v};
        for (int j = 0; j < 3; ++j) {
            for (size_t i = 0; i < b.ne; ++i) {
                t[i] = f32_to_bf16((*src[j])[i]);
                (*dst[j])[i] = bf16_to_f32(t[i]);
            }
            HIP_OK(hipMemcpy(dev[j], t.data(), b.ne * 2, hipMemcpyHostToDevice));
        }
    } else {
        HIP_OK(hipMemcpy(b.q, hq.data(), b.ne * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b.k, hk.data(), b.ne * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b.v, hv.data(), b.ne * 4, hipMemcpyHostToDevice));
    }
    if (want_f32_copy) {
        HIP_OK(hipMalloc(&b.qf, b.ne * 4));
        HIP_OK(hipMalloc(&b.kf, b.ne * 4));
        HIP_OK(hipMalloc(&b.vf, b.ne * 4));
        HIP_OK(hipMalloc(&b.of, b.ne * 4));
        HIP_OK(hipMemcpy(b.qf, rq.data(), b.ne * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b.kf, rk.data(), b.ne * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(b.vf, rv.data(), b.ne * 4, hipMemcpyHostToDevice));
    }
}

static std::vector<float> download(const Args& a, const void* dev, size_t ne)
{
    std::vector<float> out(ne);
    if (a.dtype == "bf16" && !a.out_f32) {
        std::vector<uint16_t> t(ne);
        HIP_OK(hipMemcpy(t.data(), dev, ne * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ne; ++i) out[i] = bf16_to_f32(t[i]);
    } else {
        HIP_OK(hipMemcpy(out.data(), dev, ne * 4, hipMemcpyDeviceToHost));
    }
    return out;
}

static int fa_ok(int rc, const char* what)
{
    if (rc != FA_OK) {
        fprintf(stderr, "%s failed (%d): %s\n", what, rc, fa_last_error());
        exit(3);
    }
    return rc;
}

static double peak_tflops(const std::string& dtype) { return dtype == "bf16" ? 2500.0 : 157.3; }

static void run_one(const Args& a, Buffers& b, int variant, const std::vector<float>* ref)
{
    const int dt = a.dtype == "bf16" ? (a.out_f32 ? FA_DTYPE_BF16_OUT_F32 : FA_DTYPE_BF16) : FA_DTYPE_F32;
    // --dtype f32: the exact fp32 kernel; --dtype f32s: fp32 tensors through the split kernel (variant = its tiling mode);
    // --kernel overrides the family
    int family = a.dtype == "f32s" ? FA_KERNEL_SPLIT : FA_KERNEL_MFMA;
    if (a.kernel == "auto") family = FA_KERNEL_AUTO;
    else if (a.kernel == "mfma") family = FA_KERNEL_MFMA;
    else if (a.kernel == "split") family = FA_KERNEL_SPLIT;
    else if (a.kernel == "p16") family = 4;     // (ablation library only: P and V in fp16, one term)
    else if (a.kernel == "p16x2") family = 5;   // (ablation library only: two fp16 terms of P)
    else if (a.kernel == "pb2") family = FA_KERNEL_PB2;
    else if (!a.kernel.empty()) {
        fprintf(stderr, "unknown --kernel %s\n", a.kernel.c_str());
        exit(2);
    }
    const int kernel = family | (variant << 8);
    HIP_OK(hipMemset(b.o, 0xff, b.ne * 4));
    // the non-allocating form of the boundary: the caller sizes and owns the scratch (most shapes need none)
    const size_t ws_bytes = fa_workspace_bytes(a.bh, a.n, a.d, a.causal, dt, kernel);
    void* ws = nullptr;
    if (ws_bytes > 0) HIP_OK(hipMalloc(&ws, ws_bytes));
    fa_ok(fa_forward_ws(b.q, b.k, b.v, b.o, nullptr, a.bh, a.n, a.d, a.scale, a.causal, dt, kernel, ws, ws_bytes, nullptr), "fa_forward_ws");
    HIP_OK(hipDeviceSynchronize());
    int32_t route = 0;
    fa_ok(fa_last_forward_route(nullptr, &route), "fa_last_forward_route");
    if (ws) HIP_OK(hipFree(ws));
    double max_err = -1.0;
    size_t n_nan = 0;
    if (ref) {
        const std::vector<float> got = download(a, b.o, b.ne);
        max_err = 0.0;
        for (size_t i = 0; i < b.ne; ++i) {
            if (std::isnan(got[i])) {
                ++n_nan;
                continue;
            }
            const double e = fabs((double)got[i] - (double)(*ref)[i]);
            if (e > max_err) max_err = e;
        }
    }
    float ms = 0.0f;
    fa_ok(fa_time_forward(b.q, b.k, b.v, b.o, a.bh, a.n, a.d, a.scale, a.causal, dt, kernel, nullptr, a.warmup, a.iters, &ms),
          "fa_time_forward");
    const double flop = (a.causal ? 2.0 : 4.0) * (double)a.bh * (double)a.n * (double)a.n * (double)a.d;
    const double tf = flop / (ms * 1e-3) / 1e12;
    printf("{\"mode\": \"%s\", \"dtype\": \"%s\", \"kernel\": \"%s\", \"out_f32\": %d, \"route\": %d, \"variant\": %d, \"bh\": %ld, \"n\": %ld, \"d\": %d, \"causal\": %d, "
           "\"scale\": %g, \"ms\": %.4f, \"tflops\": %.2f, \"frac_mfma_peak\": %.4f, \"max_abs_err_vs_naive\": %.3e, "
           "\"nan\": %zu, \"iters\": %d}\n",
           a.mode.c_str(), a.dtype.c_str(), a.kernel.empty() ? "default" : a.kernel.c_str(), a.out_f32, (int)route, variant, a.bh, a.n, a.d, a.causal,
           (double)a.scale, ms, tf,
           tf / peak_tflops(a.dtype), max_err, n_nan, a.iters);
    fflush(stdout);
}

// llm.c's make_random_float (src/llm.c/common.h): rand() / RAND_MAX * 2 - 1 after srand(0)
static int run_llmc(const Args& a)
{
    const int B = a.B, T = a.T, C = a.C, NH = a.NH;
    if (B < 1 || T < 1 || C < 1 || NH < 1 || C % NH != 0) {
        fprintf(stderr, "bad llmc shape\n");
        return 2;
    }
    const int hs = C / NH;
    const size_t n_inp = (size_t)B * T * 3 * C, n_out = (size_t)B * T * C;
    std::vector<float> inp(n_inp);
    srand(0);
    for (size_t i = 0; i < n_inp; ++i) inp[i] = ((float)rand() / (float)RAND_MAX) * 2.0f - 1.0f;
    float *d_inp = nullptr, *d_out = nullptr;
    HIP_OK(hipMalloc(&d_inp, n_inp * 4));
    HIP_OK(hipMalloc(&d_out, n_out * 4));
    HIP_OK(hipMemcpy(d_inp, inp.data(), n_inp * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_out, 0xff, n_out * 4));
    fa_ok(fa_forward_packed_qkv(d_inp, d_out, B, T, C, NH, nullptr), "fa_forward_packed_qkv");
    HIP_OK(hipDeviceSynchronize());
    int32_t route = 0;
    fa_ok(fa_last_forward_route(nullptr, &route), "fa_last_forward_route");
    std::vector<float> got(n_out);
    HIP_OK(hipMemcpy(got.data(), d_out, n_out * 4, hipMemcpyDeviceToHost));

    // rung 0 on the same values, plain (B*NH, T, hs) layout (the permute_kernel of the reference, :519-545, done on the host)
    const size_t ne = (size_t)B * NH * T * hs;
    std::vector<float> q(ne), k(ne), v(ne);
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int h = 0; h < NH; ++h) {
                const float* row = inp.data() + ((size_t)b * T + t) * 3 * C + (size_t)h * hs;
                const size_t dst = (((size_t)b * NH + h) * T + t) * hs;
                memcpy(&q[dst], row, hs * 4);
                memcpy(&k[dst], row + C, hs * 4);
                memcpy(&v[dst], row + 2 * C, hs * 4);
            }
    float *dq, *dk, *dv, *dref;
    HIP_OK(hipMalloc(&dq, ne * 4));
    HIP_OK(hipMalloc(&dk, ne * 4));
    HIP_OK(hipMalloc(&dv, ne * 4));
    HIP_OK(hipMalloc(&dref, ne * 4));
    HIP_OK(hipMemcpy(dq, q.data(), ne * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dk, k.data(), ne * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dv, v.data(), ne * 4, hipMemcpyHostToDevice));
    fa_ok(fa_forward_ex(dq, dk, dv, dref, nullptr, (int64_t)B * NH, T, hs, 1.0f / sqrtf((float)hs), 1, FA_DTYPE_F32, FA_KERNEL_NAIVE, nullptr), "naive");
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> ref(ne);
    HIP_OK(hipMemcpy(ref.data(), dref, ne * 4, hipMemcpyDeviceToHost));
    double max_err = 0.0;
    size_t n_bad = 0;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < T; ++t)
            for (int h = 0; h < NH; ++h)
                for (int c = 0; c < hs; ++c) {
                    const float g = got[((size_t)b * T + t) * C + (size_t)h * hs + c];
                    const float r = ref[(((size_t)b * NH + h) * T + t) * hs + c];
                    const double e = fabs((double)g - (double)r);
                    if (!(e <= 1e-4)) ++n_bad;   // validate_result tolerance (:1262); NaN counts as bad
                    if (e > max_err) max_err = e;
                }
    // benchmark_kernel: `repeats` launches between two events on the null stream, mean
    const int repeats = a.iters_given ? a.iters : 100;
    for (int i = 0; i < 10; ++i) fa_ok(fa_forward_packed_qkv(d_inp, d_out, B, T, C, NH, nullptr), "warm-up");
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < repeats; ++i) fa_ok(fa_forward_packed_qkv(d_inp, d_out, B, T, C, NH, nullptr), "fa_forward_packed_qkv");
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)repeats;
    const double flop = 2.0 * (double)B * NH * (double)T * (double)T * hs;   // causal
    printf("{\"mode\": \"llmc\", \"B\": %d, \"T\": %d, \"C\": %d, \"NH\": %d, \"route\": %d, \"max_abs_err_vs_naive\": %.3e, \"tol\": 1e-4, "
           "\"not_within_tol\": %zu, \"pass\": %s, \"repeats\": %d, \"ms\": %.4f, \"tflops\": %.2f}\n",
           B, T, C, NH, (int)route, max_err, n_bad, n_bad == 0 ? "true" : "false", repeats, ms, flop / (ms * 1e-3) / 1e12);
    return n_bad == 0 ? 0 : 1;
}

// BASELINE config 5 (B = 64, H = 16 -> 1024 slabs, bf16) over N shards: contiguous split of the batch*head axis, no collective
static int run_sharded(const Args& a)
{
    int ndev = fa_device_count();
    if (ndev < 1) {
        fprintf(stderr, "no device\n");
        return 4;
    }
    const int ns = a.devices > 0 ? a.devices : ndev;
    const bool bf = a.dtype == "bf16";
    const int dt = bf ? FA_DTYPE_BF16 : FA_DTYPE_F32;
    const size_t esz = bf ? 2 : 4;
    std::vector<int32_t> devs(ns);
    std::vector<int64_t> bhs(ns);
    std::vector<void*> q(ns, nullptr), k(ns, nullptr), v(ns, nullptr), o(ns, nullptr), streams(ns, nullptr);
    std::vector<hipEvent_t> e0(ns), e1(ns);
    const int64_t base = a.bh / ns, rem = a.bh % ns;
    for (int i = 0; i < ns; ++i) {
        devs[i] = i % ndev;
        bhs[i] = base + (i < rem ? 1 : 0);
        HIP_OK(hipSetDevice(devs[i]));
        const size_t ne = (size_t)bhs[i] * a.n * a.d;
        hipStream_t st;
        HIP_OK(hipStreamCreate(&st));
        streams[i] = st;
        HIP_OK(hipEventCreate(&e0[i]));
        HIP_OK(hipEventCreate(&e1[i]));
        if (ne == 0) continue;
        HIP_OK(hipMalloc(&q[i], ne * esz));
        HIP_OK(hipMalloc(&k[i], ne * esz));
        HIP_OK(hipMalloc(&v[i], ne * esz));
        HIP_OK(hipMalloc(&o[i], ne * esz));
        // random data of the right kind without a host pass over 1 GiB per tensor: one seeded slab, repeated
        const size_t slab = (size_t)a.n * a.d;
        std::vector<float> h(slab);
        std::vector<uint16_t> hb(slab);
        void* dst[3] = {q[i], k[i], v[i]};
        for (int t = 0; t < 3; ++t) {
            Rng rng(1234u + 17u * (unsigned)t + 101u * (unsigned)i);
            for (size_t j = 0; j < slab; ++j) h[j] = rng.normal();
            if (bf) for (size_t j = 0; j < slab; ++j) hb[j] = f32_to_bf16(h[j]);
            for (int64_t s = 0; s < bhs[i]; ++s)
                HIP_OK(hipMemcpy((char*)dst[t] + (size_t)s * slab * esz, bf ? (const void*)hb.data() : (const void*)h.data(), slab * esz, hipMemcpyHostToDevice));
        }
    }
    auto once = [&]() {
        fa_ok(fa_forward_sharded(ns, devs.data(), q.data(), k.data(), v.data(), o.data(), bhs.data(), a.n, a.d, a.scale, a.causal, dt, streams.data()),
              "fa_forward_sharded");
    };
    for (int w = 0; w < (a.warmup > 0 ? a.warmup : 1); ++w) once();
    for (int i = 0; i < ns; ++i) {
        HIP_OK(hipSetDevice(devs[i]));
        HIP_OK(hipStreamSynchronize((hipStream_t)streams[i]));
    }
    for (int i = 0; i < ns; ++i) {
        HIP_OK(hipSetDevice(devs[i]));
        HIP_OK(hipEventRecord(e0[i], (hipStream_t)streams[i]));
    }
    const int iters = a.iters_given ? a.iters : 5;
    for (int it = 0; it < iters; ++it) once();
    double worst = 0.0;
    std::string per = "[";
    for (int i = 0; i < ns; ++i) {
        HIP_OK(hipSetDevice(devs[i]));
        HIP_OK(hipEventRecord(e1[i], (hipStream_t)streams[i]));
        HIP_OK(hipEventSynchronize(e1[i]));
        float ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&ms, e0[i], e1[i]));
        const double m = ms / iters;
        worst = m > worst ? m : worst;
        char buf[64];
        snprintf(buf, sizeof(buf), "%s%.4f", i ? ", " : "", m);
        per += buf;
    }
    per += "]";
    const double flop = (a.causal ? 2.0 : 4.0) * (double)a.bh * (double)a.n * (double)a.n * (double)a.d;
    printf("{\"mode\": \"sharded\", \"dtype\": \"%s\", \"shards\": %d, \"visible_devices\": %d, \"bh\": %ld, \"n\": %ld, \"d\": %d, \"causal\": %d, "
           "\"per_shard_ms\": %s, \"ms\": %.4f, \"tflops\": %.2f, \"iters\": %d}\n",
           a.dtype.c_str(), ns, ndev, a.bh, a.n, a.d, a.causal, per.c_str(), worst, flop / (worst * 1e-3) / 1e12, iters);
    return 0;
}

int main(int argc, char** argv)
{
    const Args a = parse(argc, argv);
    if (fa_device_count() < 1) {
        fprintf(stderr, "no HIP device visible\n");
        return 4;
    }
    HIP_OK(hipSetDevice(0));
    if (a.mode == "llmc") return run_llmc(a);
    if (a.mode == "sharded") return run_sharded(a);
    const size_t ne = (size_t)a.bh * a.n * a.d;
    std::vector<float> hq(ne), hk(ne), hv(ne);

    if (a.mode == "kat") {
        // test.cu:615-631: K = Q = iota (per element index), V = 1  ->  O == 1 exactly
        for (size_t i = 0; i < ne; ++i) hq[i] = hk[i] = (float)i;
        for (size_t i = 0; i < ne; ++i) hv[i] = 1.0f;
        Buffers b;
        upload(a, hq, hk, hv, b, false);
        const int dt = a.dtype == "bf16" ? FA_DTYPE_BF16 : FA_DTYPE_F32;
        float ms = 0.0f;
        fa_ok(fa_time_forward(b.q, b.k, b.v, b.o, a.bh, a.n, a.d, a.scale, a.causal, dt, FA_KERNEL_AUTO, nullptr, 0, 1, &ms),
              "fa_time_forward");
        const std::vector<float> got = download(a, b.o, ne);
        size_t bad = 0;
        for (size_t i = 0; i < ne; ++i)
            if (!(got[i] == 1.0f)) ++bad;
        printf("{\"mode\": \"kat\", \"dtype\": \"%s\", \"bh\": %ld, \"n\": %ld, \"d\": %d, \"causal\": %d, \"ms\": %.4f, "
               "\"not_one\": %zu, \"pass\": %s}\n",
               a.dtype.c_str(), a.bh, a.n, a.d, a.causal, ms, bad, bad == 0 ? "true" : "false");
        return bad == 0 ? 0 : 1;
    }

    Rng rng(0);
    for (size_t i = 0; i < ne; ++i) hq[i] = rng.normal();
    for (size_t i = 0; i < ne; ++i) hk[i] = rng.normal();
    for (size_t i = 0; i < ne; ++i) hv[i] = rng.normal();
    Buffers b;
    upload(a, hq, hk, hv, b, a.check != 0);

    std::vector<float> ref;
    if (a.check) {
        // on-device cross-check: the rung-0 fp32 kernel on the exact values the fast kernel consumes
        fa_ok(fa_forward_ex(b.qf, b.kf, b.vf, b.of, nullptr, a.bh, a.n, a.d, a.scale, a.causal, FA_DTYPE_F32, FA_KERNEL_NAIVE, nullptr),
              "naive");
        HIP_OK(hipDeviceSynchronize());
        ref.resize(ne);
        HIP_OK(hipMemcpy(ref.data(), b.of, ne * 4, hipMemcpyDeviceToHost));
    }
    if (a.mode == "prof") {
        // in-kernel phase timers of the pp3 main loop (variant 22 writes 8 floats per wave into the lse buffer)
        float* lse = nullptr;
        HIP_OK(hipMalloc(&lse, (size_t)a.bh * a.n * 4));
        HIP_OK(hipMemset(lse, 0, (size_t)a.bh * a.n * 4));
        fa_ok(fa_forward_ex(b.q, b.k, b.v, b.o, lse, a.bh, a.n, a.d, a.scale, 0, FA_DTYPE_BF16, FA_KERNEL_MFMA | (22 << 8), nullptr), "prof");
        HIP_OK(hipDeviceSynchronize());
        const size_t nw = (size_t)a.bh * ((a.n + 255) / 256) * 4;
        std::vector<float> h(nw * 16);
        HIP_OK(hipMemcpy(h.data(), lse, h.size() * 4, hipMemcpyDeviceToHost));
        double acc[9] = {0};
        for (size_t w = 0; w < nw; ++w)
            for (int i = 0; i < 9; ++i) acc[i] += h[w * 16 + i];
        const double steps = acc[8] / nw * 2.0;
        printf("pp3 phase profile (mean cycles per wave; %zu waves, %.0f steps each)\n", nw, steps);
        const char* nm[8] = {"Q  phase (QK MFMAs || exp A)", "P1 phase (PV A || exp B)", "P2 phase (PV B || max)", "rescale decision",
                             "stage top: wait for own DMA", "stage top: barrier", "stage top: DMA issue", "whole fast loop"};
        for (int i = 0; i < 8; ++i)
            printf("  %-38s total %10.0f   per step %8.1f\n", nm[i], acc[i] / nw, acc[i] / nw / steps);
        double mn = 1e30, mx = 0, av = 0;
        for (size_t w = 0; w < nw; ++w) { const double x = h[w * 16 + 9]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; av += x / nw; }
        printf("  whole kernel per wave: min %.0f mean %.0f max %.0f cycles\n", mn, av, mx);
        return 0;
    }
    if (a.mode == "timeline") {
        // where and when every workgroup of a causal NB = 2 launch ran (ablation library, variant 71: the product's launch order with stamps):
        // one line per workgroup -- block, xcc, se, cu, q tile, slab, 64-key stages, entry and exit in 10 ns ticks from the first entry
        float* lse = nullptr;
        HIP_OK(hipMalloc(&lse, (size_t)a.bh * a.n * 4));
        for (int rep = 0; rep < 30; ++rep)   // warm clocks
            fa_ok(fa_forward_ex(b.q, b.k, b.v, b.o, lse, a.bh, a.n, a.d, a.scale, a.causal, FA_DTYPE_BF16, FA_KERNEL_MFMA | (a.variant << 8), nullptr), "timeline");
        HIP_OK(hipMemset(lse, 0, (size_t)a.bh * a.n * 4));
        fa_ok(fa_forward_ex(b.q, b.k, b.v, b.o, lse, a.bh, a.n, a.d, a.scale, a.causal, FA_DTYPE_BF16, FA_KERNEL_MFMA | (a.variant << 8), nullptr), "timeline");
        HIP_OK(hipDeviceSynchronize());
        const size_t nwg = (size_t)a.bh * ((a.n + 255) / 256);
        std::vector<unsigned> h(nwg * 4 * 8);
        HIP_OK(hipMemcpy(h.data(), lse, h.size() * 4, hipMemcpyDeviceToHost));
        unsigned t0 = 0xffffffffu;
        for (size_t g = 0; g < nwg; ++g) t0 = h[g * 32] < t0 ? h[g * 32] : t0;
        printf("# block xcc se cu qtile slab stages t_in t_out   (wave 0 of each workgroup; 10 ns ticks)\n");
        for (size_t g = 0; g < nwg; ++g) {
            const unsigned* u = &h[g * 32];
            unsigned t_out = 0;
            for (int wv = 0; wv < 4; ++wv) t_out = h[g * 32 + wv * 8 + 1] > t_out ? h[g * 32 + wv * 8 + 1] : t_out;
            // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
            printf("%u %u %u %u %u %u %u %u %u\n", u[6], u[3] & 15u, (u[2] >> 13) & 7u, (u[2] >> 8) & 15u, u[4], u[5], u[7], u[0] - t0, t_out - t0);
        }
        return 0;
    }
    if (a.mode == "prof4") {
        // cycle stamps around the fast loop of the NB = 4 kernel (ablation library; variants 60..68 write 4 floats per wave to the lse buffer)
        float* lse = nullptr;
        HIP_OK(hipMalloc(&lse, (size_t)a.bh * a.n * 4));
        // (--dtype f32: the fp32 default's pipelined pass, a library built with -DFA_SPLIT_STAMPS=1; --variant = its tiling, 3: 128-row, 4: 256-row workgroups)
        const bool f32 = a.dtype == "f32" || a.dtype == "f32s";
        const int p4_dtype = f32 ? FA_DTYPE_F32 : FA_DTYPE_BF16, p4_kernel = (f32 ? FA_KERNEL_SPLIT : FA_KERNEL_MFMA) | (a.variant << 8);
        for (int rep = 0; rep < 30; ++rep)   // warm clocks
            fa_ok(fa_forward_ex(b.q, b.k, b.v, b.o, lse, a.bh, a.n, a.d, a.scale, a.causal, p4_dtype, p4_kernel, nullptr), "prof4");
        HIP_OK(hipMemset(lse, 0, (size_t)a.bh * a.n * 4));
        fa_ok(fa_forward_ex(b.q, b.k, b.v, b.o, lse, a.bh, a.n, a.d, a.scale, a.causal, p4_dtype, p4_kernel, nullptr), "prof4");
        HIP_OK(hipDeviceSynchronize());
        const int rows_per_wg = f32 ? (a.variant == 3 ? 128 : 256) : a.variant == 70 ? 256 : 512;
        const size_t nw = (size_t)a.bh * ((a.n + rows_per_wg - 1) / rows_per_wg) * 4;
        std::vector<float> h(nw * 8);
        HIP_OK(hipMemcpy(h.data(), lse, h.size() * 4, hipMemcpyDeviceToHost));
        double cyc = 0, real = 0, steps = 0, mx = 0, whole = 0, whole_mx = 0, ph[4] = {0, 0, 0, 0};
        for (size_t w = 0; w < nw; ++w) {
            cyc += h[w * 8]; real += h[w * 8 + 1]; steps += h[w * 8 + 2]; mx = h[w * 8] > mx ? h[w * 8] : mx;
            whole += h[w * 8 + 3]; whole_mx = h[w * 8 + 3] > whole_mx ? h[w * 8 + 3] : whole_mx;
            for (int i = 0; i < 4; ++i) ph[i] += h[w * 8 + 4 + i];
        }
        printf("{\"mode\": \"prof4\", \"variant\": %d, \"waves\": %zu, \"steps_per_wave\": %.0f, \"cycles_per_step\": %.1f, \"max_wave_cycles_per_step\": %.1f, "
               "\"loop_us\": %.2f, \"shader_mhz\": %.0f, \"tile_cycles_mean\": %.0f, \"tile_cycles_max\": %.0f, \"outside_loop_cycles_mean\": %.0f, "
               "\"inputs_landed\": %.0f, \"first_scores\": %.0f, \"tail_stages\": %.0f, \"epilogue_issue\": %.0f, \"stores_landing\": %.0f}\n", a.variant, nw, steps / nw,
               cyc / steps, mx / (steps / nw), real / nw / 100.0, cyc / real * 100.0, whole / nw, whole_mx, (whole - cyc) / nw,
               ph[0] / nw, ph[1] / nw, ph[2] / nw, ph[3] / nw, (whole - cyc - ph[0] - ph[1] - ph[2] - ph[3]) / nw);
        {   // spread of the per-wave loop time (two workgroups sharing a CU: who gets the issue slots?)
            std::vector<float> per(nw), life(nw);
            for (size_t w = 0; w < nw; ++w) per[w] = h[w * 8] / (h[w * 8 + 2] > 0 ? h[w * 8 + 2] : 1.0f), life[w] = h[w * 8 + 3];
            std::sort(per.begin(), per.end());
            std::sort(life.begin(), life.end());
            printf("#  loop cycles per step, percentiles 0 10 25 50 75 90 100: ");
            for (double q : {0.0, 0.10, 0.25, 0.50, 0.75, 0.90, 1.0}) printf("%.0f ", per[(size_t)(q * (nw - 1))]);
            printf("\n#  tile cycles, the same percentiles: ");
            for (double q : {0.0, 0.10, 0.25, 0.50, 0.75, 0.90, 1.0}) printf("%.0f ", life[(size_t)(q * (nw - 1))]);
            printf("\n");
        }
        if (a.causal) {   // per q tile of slab 0 (tiles are launched heaviest first: tile index = q_tiles - 1 - launch position)
            const size_t qt = (a.n + rows_per_wg - 1) / rows_per_wg;
            printf("# slab 0, per launch position: steps, loop cycles, inputs_landed, first_scores, tail, epilogue_issue, whole tile\n");
            for (size_t t = 0; t < qt; t += (qt > 16 ? qt / 16 : 1)) {
                for (int wv = 0; wv < 4; wv += 3) {     // wave 0 (lowest rows: fewest diagonal sub-tiles) and wave 3 (highest rows: most)
                    const float* r = &h[(t * 4 + wv) * 8];   // workgroup t (xcd_remap keeps slab 0's tiles in the first positions of XCD 0)
                    printf("#  pos %3zu wave %d: %5.0f %8.0f %7.0f %7.0f %7.0f %7.0f %8.0f\n", t, wv, r[2], r[0], r[4], r[5], r[6], r[7], r[3]);
                }
            }
        }
        return 0;
    }
    if (a.mode == "sweep") {
        const int nvar = a.dtype == "bf16" ? 9 : 1;
        for (int v = 0; v < nvar; ++v) run_one(a, b, v, a.check ? &ref : nullptr);
    } else {
        run_one(a, b, a.variant, a.check ? &ref : nullptr);
    }
    return 0;
}
