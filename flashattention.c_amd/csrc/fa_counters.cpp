// fa_counters.cpp -- the report ring of fp32 FA_KERNEL_AUTO forwards, the kernels' slow-path counters, the private scratch pools of the
// convenience entry points and the process-wide host counters behind fa_get_stats() (fa_host.h).
//
// Rounds 2-5 kept the 32-bit report word of a forward in per-device slot tables (one slot per stream, LRU hand-over behind per-slot events,
// capture slots tied to their graph by a hipUserObject, a mutex held while a forward was enqueued).  Since the fp32 fallback moved INSIDE
// the kernel (round 4) the word only reports, so round 6 replaced all of that by a ring: word `serial % kReportRing`, raised = equal to the
// call's serial.  Nothing to lock, nothing to release, nothing to clear; a captured forward takes no word (its replays would share one).
#include "fa_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace fa_host {

__device__ uint32_t g_report_ring[kReportRing];
__device__ unsigned long long g_stat_ring[kReportRing][2];   // pre-pass maxima of the ablation library's t3 chain, tagged with the call's serial
__device__ unsigned long long g_cliff_words[2];              // FwdParams::cliffs: tiles redone, workgroups redone in fp32 arithmetic
namespace {
std::atomic<uint32_t*> g_ring_base[kMaxDevices];
std::atomic<unsigned long long*> g_stat_base[kMaxDevices];
std::atomic<unsigned long long*> g_cliff_base[kMaxDevices];
std::atomic<uint32_t> g_serial{1};
}  // namespace

Stats g_stats;
thread_local ReportRef t_last_report;
thread_local int t_last_chain = 0;

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}

bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// address of a __device__ symbol on the current device, looked up once per device (nothing to allocate: legal while capturing)
template <class T>
static T* device_symbol(std::atomic<T*> (&cache)[kMaxDevices], const void* symbol)
{
    const int dev = current_device();
    if (dev < 0) return nullptr;
    T* base = cache[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, symbol) != hipSuccess || sym == nullptr) {
            (void)hipGetLastError();
            return nullptr;
        }
        base = static_cast<T*>(sym);
        cache[dev].store(base, std::memory_order_release);
    }
    return base;
}

bool next_report(ReportRef& r)
{
    uint32_t* ring = device_symbol(g_ring_base, HIP_SYMBOL(g_report_ring));
    unsigned long long* stats = device_symbol(g_stat_base, HIP_SYMBOL(g_stat_ring));
    if (ring == nullptr || stats == nullptr) return false;
    uint32_t serial = g_serial.fetch_add(1, std::memory_order_relaxed);
    if (serial == 0) serial = g_serial.fetch_add(1, std::memory_order_relaxed);   // 0 is the ring's initial content
    r.word = ring + serial % kReportRing;
    r.serial = serial;
    r.stats = stats + 2 * (size_t)(serial % kReportRing);
    return true;
}

// One pair of counter words per GPU, bumped by the kernels with DEVICE-scope atomics on their slow paths only.  (The first form of round 5
// kept them in pinned host memory behind system-scope atomics: every redone tile then paid a serialised PCIe round trip, ~1 us each --
// profiles/r05_exp/exp9_redo_cost.py.)
unsigned long long* cliff_counters() { return device_symbol(g_cliff_base, HIP_SYMBOL(g_cliff_words)); }

// sum over the devices this process has launched on; a blocking 8-byte copy per device
unsigned long long cliff_count(int which)
{
    unsigned long long total = 0;
    for (int dev = 0; dev < kMaxDevices; ++dev) {
        const unsigned long long* base = g_cliff_base[dev].load(std::memory_order_acquire);
        if (base == nullptr) continue;
        unsigned long long w = 0;
        if (hipMemcpy(&w, base + which, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess) total += w;
        else (void)hipGetLastError();
    }
    return total;
}

// ---- scratch ------------------------------------------------------------------------------------------------------------------
// The C ABI proper never allocates: fa_forward_ws runs in a caller-owned workspace whose size fa_workspace_bytes reports.  The
// convenience entry points (fa_forward, fa_forward_ex, the sharded and timing entries) take the same bytes from a PRIVATE
// stream-ordered pool per device (hipMemPoolCreate; its release threshold is ours to raise -- the device's default pool, which the
// host application and torch may be using, is never touched) and return them behind the last kernel that reads them.
struct DevicePool {
    std::atomic<int> state{0};   // 0 = untried, 1 = being created, 2 = ready, 3 = unavailable (plain hipMallocAsync then)
    hipMemPool_t pool = nullptr;
};
DevicePool g_pools[kMaxDevices];

hipMemPool_t private_pool(int dev)
{
    if (dev < 0) return nullptr;
    DevicePool& dp = g_pools[dev];
    int st = dp.state.load(std::memory_order_acquire);
    if (st == 0) {
        int expect = 0;
        if (dp.state.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
            hipMemPoolProps props;
            memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t pool = nullptr;
            if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool != nullptr) {
                uint64_t keep = ~0ull;   // keep what steady-state calls hand back: they then never reach the driver
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
                dp.pool = pool;
                dp.state.store(2, std::memory_order_release);
            } else {
                (void)hipGetLastError();
                dp.state.store(3, std::memory_order_release);
            }
        }
        st = dp.state.load(std::memory_order_acquire);
    }
    while (st == 1) st = dp.state.load(std::memory_order_acquire);
    return st == 2 ? dp.pool : nullptr;
}

hipError_t scratch_alloc(void** ptr, size_t bytes, hipStream_t stream)
{
    hipMemPool_t pool = private_pool(current_device());
    if (pool != nullptr) return hipMallocFromPoolAsync(ptr, bytes, pool, stream);
    return hipMallocAsync(ptr, bytes, stream);
}

}  // namespace fa_host
