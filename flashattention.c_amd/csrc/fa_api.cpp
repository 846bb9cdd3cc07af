// fa_api.cpp -- the extern "C" boundary declared in include/flashattn_amd.h, and nothing else: every entry point validates, builds the
// parameter block and hands over to fa_host:: (fa_host.h lists the translation units behind it).
//
// Host-side counterpart of forward() + run_flash_tiled_coarse{,_causal}
// (/root/reference/src/flashattention.cu:590-617).  Unlike the reference it never allocates, never synchronises (except fa_time_forward)
// and reports errors by return code + thread-local message instead of assert().
#include "fa_host.h"

#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

using namespace fa_host;

extern "C" {

int fa_forward_ex(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d, float scale,
                  int32_t causal, int32_t dtype, int32_t kernel, void* stream)
{
    g_err[0] = 0;
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    const fa::FwdParams p = make_params(q, k, v, o, lse, bh, n, d, scale);
    return launch(p, d, causal, dtype, kernel, static_cast<hipStream_t>(stream));
}

int fa_forward(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
               int32_t dtype, void* stream)
{
    return fa_forward_ex(q, k, v, o, nullptr, bh, n, d, scale, causal, dtype, FA_KERNEL_AUTO, stream);
}

size_t fa_workspace_bytes(int64_t bh, int64_t n, int32_t d, int32_t causal, int32_t dtype, int32_t kernel)
{
    if (bh < 1 || n < 1 || bh > 0x7fffffffLL || n > (1LL << 24) || d < 1) return 0;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_BF16_OUT_F32) return 0;
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(g_err));   // a size query does not disturb fa_last_error()
    const fa::FwdParams p = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
    const Plan pl = make_plan(p, d, causal, dtype, kernel, true);
    memcpy(g_err, keep, sizeof(g_err));
    return pl.status == FA_OK ? pl.total : 0;
}

int fa_forward_ws(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d, float scale,
                  int32_t causal, int32_t dtype, int32_t kernel, void* workspace, size_t workspace_bytes, void* stream)
{
    g_err[0] = 0;
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    if (workspace != nullptr && workspace_bytes > 0) {   // the workspace is written by the launch: it must not overlap a tensor of the call
        const uint64_t elems = (uint64_t)bh * (uint64_t)n * (uint64_t)d;
        const uint64_t in_bytes = elems * (dtype == FA_DTYPE_F32 ? 4u : 2u), out_bytes = elems * (dtype == FA_DTYPE_BF16 ? 2u : 4u);
        const uintptr_t wb = reinterpret_cast<uintptr_t>(workspace);
        const struct { const void* t; uint64_t bytes; } tensors[] = {{q, in_bytes}, {k, in_bytes}, {v, in_bytes}, {o, out_bytes}};
        for (const auto& t : tensors) {
            const uintptr_t tb = reinterpret_cast<uintptr_t>(t.t);
            if (wb < tb + t.bytes && tb < wb + workspace_bytes) return fail(FA_ERR_INVALID_ARGUMENT, "workspace overlaps a tensor of the call");
        }
    }
    const fa::FwdParams p = make_params(q, k, v, o, lse, bh, n, d, scale);
    return launch(p, d, causal, dtype, kernel, static_cast<hipStream_t>(stream), workspace, workspace_bytes, true);
}

int fa_forward_sharded_ex(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                          void* const* o, float* const* lse, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                          int32_t kernel, void* const* workspaces, const size_t* workspace_bytes, void* const* streams)
{
    return forward_sharded(n_shards, device_ids, q, k, v, o, lse, bh, n, d, scale, causal, dtype, kernel, workspaces, workspace_bytes, streams);
}

int fa_forward_sharded(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                       void* const* o, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                       void* const* streams)
{
    return fa_forward_sharded_ex(n_shards, device_ids, q, k, v, o, nullptr, bh, n, d, scale, causal, dtype, FA_KERNEL_AUTO, nullptr, nullptr, streams);
}

int fa_forward_packed_qkv(const float* inp, float* out, int32_t B, int32_t T, int32_t C, int32_t NH, void* stream)
{
    g_err[0] = 0;
    if (!inp || !out) return fail(FA_ERR_INVALID_ARGUMENT, "null pointer");
    if (B < 1 || T < 1 || C < 1 || NH < 1 || C % NH != 0)
        return fail(FA_ERR_INVALID_ARGUMENT, "bad shape B=%d T=%d C=%d NH=%d", B, T, C, NH);
    const int hs = C / NH;
    if (!head_dim_naive(hs)) return fail(FA_ERR_UNSUPPORTED, "head size %d not supported (1 .. 256)", hs);
    if (!aligned16(inp) || !aligned16(out)) return fail(FA_ERR_INVALID_ARGUMENT, "buffers must be 16-byte aligned");
    if ((int64_t)B * NH > 0x7fffffffLL) return fail(FA_ERR_INVALID_ARGUMENT, "B*NH too large");
    // (B, T, 3C): q at column h*hs, k at C + h*hs, v at 2C + h*hs of each token row
    // (attention_forward_cpu, /root/reference/src/llm.c/attention_forward.cu:66,74,115)
    fa::FwdParams p{};
    memset(&p, 0, sizeof(p));
    p.q = inp;
    p.k = inp + C;
    p.v = inp + 2 * (int64_t)C;
    p.o = out;
    p.lse = nullptr;
    p.q_batch_stride = p.kv_batch_stride = (int64_t)T * 3 * C;
    p.o_batch_stride = (int64_t)T * C;
    p.q_row_stride = p.kv_row_stride = 3 * C;
    p.o_row_stride = C;
    p.q_head_stride = p.kv_head_stride = p.o_head_stride = hs;
    p.heads = NH;
    p.n = T;
    p.bh = B * NH;
    p.scale = 1.0f / sqrtf((float)hs);  // attention_forward.cu:61,1123
    p.scale_log2e = p.scale * fa::kLog2e;
    return launch(p, hs, /*causal=*/1, FA_DTYPE_F32, FA_KERNEL_AUTO, static_cast<hipStream_t>(stream));
}

int fa_time_forward(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                    int32_t causal, int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters, float* ms_per_forward)
{
    const char* env = getenv("FA_TIME_GRAPH");  // experiment switch of the C driver
    return time_forward_impl(q, k, v, o, bh, n, d, scale, causal, dtype, kernel, stream, warmup, iters, ms_per_forward,
                             env && env[0] == '1');
}

int fa_time_forward_graph(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                          int32_t causal, int32_t dtype, int32_t kernel, int32_t warmup, int32_t iters, float* ms_per_forward)
{
    return time_forward_impl(q, k, v, o, bh, n, d, scale, causal, dtype, kernel, nullptr, warmup, iters, ms_per_forward, true);
}

int fa_last_forward_route(void* stream, int32_t* route)
{
    g_err[0] = 0;
    if (!route) return fail(FA_ERR_INVALID_ARGUMENT, "null route pointer");
    *route = 0;
    if (t_last_chain == 0) return FA_OK;
    uint32_t word = 0;
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemcpy(&word, t_last_report.word, sizeof(word), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "reading the forward's report word failed: %s", hipGetErrorString(e));
    *route = word == t_last_report.serial ? 2 : 1;
    return FA_OK;
}

int fa_get_stats(fa_stats* out, size_t struct_bytes)
{
    if (!out || struct_bytes < sizeof(uint64_t)) return fail(FA_ERR_INVALID_ARGUMENT, "null stats pointer or struct_bytes below one field");
    fa_stats st;
    memset(&st, 0, sizeof(st));
    st.struct_bytes = sizeof(st);
    st.forwards = g_stats.forwards.load(std::memory_order_relaxed);
    st.scratch_replans = g_stats.scratch_replans.load(std::memory_order_relaxed);
    memcpy(out, &st, struct_bytes < sizeof(st) ? struct_bytes : sizeof(st));   // a caller built against a shorter struct gets its prefix
    return FA_OK;
}

int fa_read_device_counters(uint64_t* tiles_redone, uint64_t* workgroups_fp32)
{
    g_err[0] = 0;
    if (tiles_redone) *tiles_redone = cliff_count(0);
    if (workgroups_fp32) *workgroups_fp32 = cliff_count(1);
    return FA_OK;
}

const char* fa_last_error(void) { return g_err; }

int fa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* fa_version(void)
{
#if FA_ABLATION
    return "flashattn_amd abi 6 gfx950 (hip, mfma f32 32x32x2 / bf16 32x32x16, lds-dma) +ablation";
#else
    return "flashattn_amd abi 6 gfx950 (hip, mfma f32 32x32x2 / bf16 32x32x16, lds-dma)";
#endif
}

const char* fa_kernel_name_for(int32_t dtype, int32_t d, int32_t causal, int64_t bh, int64_t n)
{
    if (bh < 1 || n < 1 || !head_dim_naive(d)) return nullptr;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_BF16_OUT_F32) return nullptr;
    if (!head_dim_supported(d)) return head_dim_exact_f32(d) ? "fa_fwd_f32_kernel" : "fa_naive_f32_kernel";
    if (dtype == FA_DTYPE_F32) {
        if (f32_auto_is_exact()) return "fa_fwd_f32_kernel";
        return "fa_fwd_f32_split_kernel";
    }
    if (dtype == FA_DTYPE_BF16) {
        const fa::FwdParams pk = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        if (keysplit_factor(pk, d, causal) > 1) return "fa_fwd_bf16_x2_kernel";       // small grids: key-split launch of the NB = 2 kernel
        return fa::bf16_kernel_name(bh, n, d, causal);
    }
    if (dtype == FA_DTYPE_BF16_OUT_F32) {   // the accurate P (see fa_dtype): hi + lo bf16 terms in the one-wave-per-SIMD kernel (slabs below 4 GiB)
        if (((n - 1) * d + d) * 2 >= 0xffffffffLL) return "fa_fwd_f32_split_kernel";
        const fa::FwdParams pk = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        if (keysplit_factor(pk, d, causal, false, true) > 1) return "fa_fwd_bf16_x2_pb2_kernel";   // small grids: key-split launch of the NB = 2 kernel
        return d == 64 && fa::bf16_pb2_uses_x4(bh, n, causal) ? "fa_fwd_bf16_x4_pb2_kernel" : "fa_fwd_bf16_x2_pb2_kernel";
    }
    return nullptr;
}

const char* fa_kernel_name(int32_t dtype, int32_t d, int32_t causal) { return fa_kernel_name_for(dtype, d, causal, 16, 8192); }

}  // extern "C"
