// fa_shard.cpp -- fa_forward_sharded: the batch*head axis split across the devices of one node, no collective (every blockIdx.x of the
// reference grid is independent: /root/reference/src/flashattention.cu:144); one persistent host thread per shard (fa_host.h).
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

// ---- fa_forward_sharded: one persistent host thread per shard index -----------------------------------------------------------------
// A forward can be a chain of launches plus a pool allocation; enqueued from one thread the last device would start a whole table's
// worth of host time behind the first.  Round 3 created and joined a std::thread per shard on every call (tens of microseconds each on
// the path of a ~0.3 ms launch: ADVICE r03); now worker i is created on first use, sleeps on a condition variable between calls and is
// never destroyed (the pool is leaked on purpose: no join at process exit, where HIP may already be gone).  One sharded call at a
// time uses the pool (g_shard_call); a second caller runs meanwhile on threads of its own, as before.
namespace {
struct ShardWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, started = false;
    std::thread th;
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job; });
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            j();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
};
struct ShardPool {
    std::mutex call_mu;                      // one sharded call at a time
    ShardWorker workers[kMaxDevices];
};
ShardPool* shard_pool()
{
    static ShardPool* pool = new ShardPool();   // never freed
    return pool;
}
}  // namespace

// run work(i) for every index in `idx`, each on its own thread, and wait for all of them
void run_on_shard_threads(const std::vector<int>& idx, const std::function<void(int)>& work)
{
    ShardPool* pool = shard_pool();
    std::unique_lock<std::mutex> call(pool->call_mu, std::try_to_lock);
    if (!call.owns_lock() || (int)idx.size() > kMaxDevices) {   // the pool is busy with another caller's shards: threads of our own
        std::vector<std::thread> th;
        for (int i : idx) {
            try {
                th.emplace_back(work, i);
            } catch (const std::exception&) {   // no thread to be had (nothing may cross the extern "C" boundary): this shard from here
                work(i);
            }
        }
        for (auto& t : th) t.join();
        return;
    }
    std::vector<int> queued;
    for (size_t k = 0; k < idx.size(); ++k) {
        ShardWorker& w = pool->workers[k];
        const int i = idx[k];
        bool ok = true;
        if (!w.started) {
            try {
                w.th = std::thread([&w] { w.loop(); });
                w.th.detach();
                w.started = true;
            } catch (const std::exception&) {
                ok = false;
            }
        }
        if (!ok) {
            work(i);
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.job = [&work, i] { work(i); };
            w.has_job = true;
            w.done = false;
        }
        w.cv.notify_all();
        queued.push_back((int)k);
    }
    for (int k : queued) {
        ShardWorker& w = pool->workers[k];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
    }
}

int forward_sharded(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                    void* const* o, float* const* lse, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                    int32_t kernel, void* const* workspaces, const size_t* workspace_bytes, void* const* streams)
{
    g_err[0] = 0;
    if (n_shards < 1 || !device_ids || !q || !k || !v || !o || !bh)
        return fail(FA_ERR_INVALID_ARGUMENT, "fa_forward_sharded: bad shard table");
    if ((workspaces == nullptr) != (workspace_bytes == nullptr))
        return fail(FA_ERR_INVALID_ARGUMENT, "fa_forward_sharded_ex: workspaces and workspace_bytes come together (both NULL: the convenience path's private pools)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FA_ERR_NO_DEVICE, "no HIP device visible");
    // a shard table that names one device twice is almost always a bug of the caller (two shards then queue up behind each other
    // instead of running side by side); FA_ALLOW_SAME_DEVICE=1 lifts the check for single-GPU test boxes
    const char* same = getenv("FA_ALLOW_SAME_DEVICE");
    const bool allow_same = same != nullptr && same[0] == '1';
    for (int i = 0; i < n_shards; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: device %d not in [0, %d)", i, device_ids[i], ndev);
        if (bh[i] < 0) return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: negative bh", i);
        if (bh[i] == 0) continue;
        if (int rc = validate_common(q[i], k[i], v[i], o[i], bh[i], n, d, scale, dtype)) return rc;
        for (int j = 0; j < i && !allow_same; ++j)
            if (bh[j] > 0 && device_ids[j] == device_ids[i])
                return fail(FA_ERR_INVALID_ARGUMENT, "shards %d and %d both name device %d (set FA_ALLOW_SAME_DEVICE=1 to allow it)", j, i, device_ids[i]);
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return fail(FA_ERR_HIP, "hipGetDevice failed");
    t_last_chain = 0;   // the shards' chains belong to their worker threads: fa_last_forward_route() of this thread reports "no chain"
    // One host thread per shard (run_on_shard_threads).  The current device is per host thread in HIP, so the workers do not disturb the
    // caller's; each worker's scratch comes from its shard's workspace, or from its own device's private pool.
    std::vector<int> rcs((size_t)n_shards, FA_OK);
    std::vector<std::string> msgs((size_t)n_shards);
    const std::function<void(int)> work = [&](int i) {
        const hipError_t e = hipSetDevice(device_ids[i]);
        if (e != hipSuccess) {
            rcs[i] = fail(FA_ERR_HIP, "hipSetDevice(%d): %s", device_ids[i], hipGetErrorString(e));
        } else {
            const fa::FwdParams p = make_params(q[i], k[i], v[i], o[i], lse ? lse[i] : nullptr, bh[i], n, d, scale);
            hipStream_t st = streams ? static_cast<hipStream_t>(streams[i]) : nullptr;
            if (workspaces != nullptr) rcs[i] = launch(p, d, causal, dtype, kernel, st, workspaces[i], workspace_bytes[i], true);
            else rcs[i] = launch(p, d, causal, dtype, kernel, st);
        }
        if (rcs[i] != FA_OK) msgs[i] = g_err;
    };
    std::vector<int> active;
    for (int i = 0; i < n_shards; ++i)
        if (bh[i] > 0) active.push_back(i);   // bh[i] == 0: more devices than slabs, this shard is empty
    if (active.size() == 1) work(active[0]);
    else if (active.size() > 1) run_on_shard_threads(active, work);
    t_last_chain = 0;
    (void)hipSetDevice(prev);
    for (int i = 0; i < n_shards; ++i)
        if (rcs[i] != FA_OK) return fail(rcs[i], "shard %d (device %d): %s", i, device_ids[i], msgs[i].c_str());
    return FA_OK;
}

}  // namespace fa_host
