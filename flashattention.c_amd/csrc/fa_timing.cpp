// fa_timing.cpp -- fa_time_forward / fa_time_forward_graph: warm-up + `iters` forwards bracketed by HIP events on the launch stream
// (the counterpart of benchmark_kernel, /root/reference/src/llm.c/common.h:108-124); blocking (fa_host.h).
#include "fa_host.h"

#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace fa_host {

// The blocking measurement entry points own a workspace for the duration of the measurement (hipMalloc / hipFree around the timed
// region, never inside it) and launch through the fa_forward_ws path: what is timed is the C ABI proper, and the launch chains that
// need scratch are legal inside the captured graph of fa_time_forward_graph.
int time_forward_impl(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                             int32_t causal, int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters,
                             float* ms_per_forward, bool graph_replay)
{
    g_err[0] = 0;
    if (!ms_per_forward || iters < 1 || warmup < 0) return fail(FA_ERR_INVALID_ARGUMENT, "bad timing arguments");
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    const fa::FwdParams p = make_params(q, k, v, o, nullptr, bh, n, d, scale);
    const Plan pl = make_plan(p, d, causal, dtype, kernel, true);
    if (pl.status != FA_OK) return pl.status;
    void* ws = nullptr;
    if (pl.total > 0 && hipMalloc(&ws, pl.total) != hipSuccess) return fail(FA_ERR_HIP, "hipMalloc(%zu) for the measurement's workspace failed", pl.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (ws) (void)hipFree(ws);
        return fail(FA_ERR_HIP, "hipEventCreate failed");
    }
    int rc = FA_OK;
    for (int i = 0; i < warmup && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s, ws, pl.total, true);
    if (rc == FA_OK && graph_replay) {  // the `iters` launches captured into one hipGraph; three replays timed one by one, the median reported
        hipStream_t cs = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(FA_ERR_HIP, "hipStreamSynchronize failed");
        if (rc == FA_OK && hipStreamCreate(&cs) != hipSuccess) rc = fail(FA_ERR_HIP, "hipStreamCreate failed");
        if (rc == FA_OK && hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal) != hipSuccess) rc = fail(FA_ERR_HIP, "begin capture failed");
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, cs, ws, pl.total, true);
        if (rc == FA_OK && hipStreamEndCapture(cs, &graph) != hipSuccess) rc = fail(FA_ERR_HIP, "end capture failed");
        if (rc == FA_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(FA_ERR_HIP, "graph instantiate failed");
        if (rc == FA_OK) {
            (void)hipGraphLaunch(exec, cs);   // warm replay
            (void)hipStreamSynchronize(cs);
            float t[3] = {0.0f, 0.0f, 0.0f};
            for (int r = 0; r < 3 && rc == FA_OK; ++r) {
                (void)hipEventRecord(e0, cs);
                (void)hipGraphLaunch(exec, cs);
                (void)hipEventRecord(e1, cs);
                const hipError_t e = hipEventSynchronize(e1);
                if (e != hipSuccess) rc = fail(FA_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
                else (void)hipEventElapsedTime(&t[r], e0, e1);
            }
            if (rc == FA_OK) {
                const float lo = fminf(fminf(t[0], t[1]), t[2]), hi = fmaxf(fmaxf(t[0], t[1]), t[2]);
                *ms_per_forward = (t[0] + t[1] + t[2] - lo - hi) / (float)iters;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (cs) (void)hipStreamDestroy(cs);
    } else if (rc == FA_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s, ws, pl.total, true);
        (void)hipEventRecord(e1, s);
        const hipError_t e = hipEventSynchronize(e1);
        if (rc == FA_OK && e != hipSuccess) rc = fail(FA_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        if (rc == FA_OK) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            *ms_per_forward = ms / (float)iters;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (ws) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(ws);
    }
    return rc;
}

}  // namespace fa_host
