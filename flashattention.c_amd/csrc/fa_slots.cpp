// fa_slots.cpp -- where the report word of a forward lives (per-device slot tables, capture slots), the private scratch pools of the
// convenience entry points, and the process-wide counters behind fa_get_stats() (fa_host.h).
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

// ---- the report word of a forward (and the ablation library's conditional launch chains) ---------------------------------------------
// Until round 4 the FA_KERNEL_AUTO path of fp32 tensors was a chain of launches in which the exact fp32 kernel, queued behind the split
// kernel, ran or skipped itself depending on a device word the split kernel raised when its logits were too wide for 16-bit operand
// terms.  Now the split kernel redoes a workgroup's rows in fp32 arithmetic itself when its operands leave the range fp16 terms hold (flag_mode 4; fa_split_kernel.h) -- ONE launch,
// and a hostile slab costs its own tiles only -- and the word merely REPORTS that some workgroup did (fa_last_forward_route).  The
// machinery below is what keeps two calls from ever sharing a word; the ablation library's chains (fp16-P kernels, the static-slot fp32
// kernel) still depend on it for correctness, the product for the truth of its report:
// "Raised" means "the word equals this call's serial number" (serials are unique per call), so a word never needs clearing between
// eager calls.  WHERE the word lives:
//   * a call that runs with a caller-owned workspace (fa_forward_ws; fa_workspace_bytes() reports at least the 256-byte header for
//     every such call) keeps its word in the first bytes of that workspace -- the caller's buffer, in use by one forward at a time
//     like every other buffer of the call;
//   * every other eager call takes the slot of its (device, stream) pair from a per-device table, and the table's mutex is held while
//     the call is enqueued: calls that share a slot are on one stream, one after the other.  An event recorded behind each call tells
//     when its slot may change hands: when the table is full the least recently used slot whose last call has COMPLETED is given to the
//     new stream (a long-running host that creates and destroys streams never runs out; round 3 handed slots out once);
//   * a call enqueued while its stream is CAPTURING takes a slot of its own and starts with a memset node that clears the word, so
//     replays of the graph report independently of each other.  The slot goes back to the table when the graph -- and every executable
//     instantiated from it -- has been destroyed (a hipUserObject retained by the capturing graph; where the runtime refuses that, the
//     slot is simply never reused).
// When no slot can be had the product launches the same kernel without a word (route 0 is reported; fa_get_stats() counts those calls);
// a chain of the ablation library launches its always-correct kernel alone.
// (Round 2 indexed a 4096-slot ring with serial % 4096: a chain whose serial was congruent -- every 4096th eager call, or a replayed
// graph -- could overwrite a raised word between the other chain's primary and its fallback kernel.)
__device__ uint32_t g_flag_ring[kFlagSlots];
__device__ unsigned long long g_stat_ring[kFlagSlots][2];   // pre-pass maxima of the t3 chain, tagged with the call's serial (experiments/fa_cvt.hip)
namespace {
std::atomic<uint32_t*> g_ring_base[kMaxDevices];
std::atomic<unsigned long long*> g_stat_base[kMaxDevices];
std::atomic<uint32_t> g_serial{1};
std::atomic<uint64_t> g_thread_numbers{1};
thread_local uint64_t t_thread_number = 0;
}  // namespace

Stats g_stats;
thread_local FlagRef t_last_flag;
thread_local int t_last_chain = 0;
thread_local int t_last_route = -1;
SlotTable g_slots[kMaxDevices];

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}

// The two slow-path counters of FwdParams::cliffs: one pair of device words per GPU (a __device__ symbol: nothing to allocate, legal while
// capturing), bumped by the kernels with DEVICE-scope atomics.  (The first form of round 5 kept them in pinned host memory behind
// system-scope atomics: every redone tile then paid a serialised PCIe round trip, ~1 us each -- a forward of 1024 redone tiles took 1 ms
// longer, ten times its own time at 128 x 1024: profiles/r05_exp/exp9_redo_cost.py.)
__device__ unsigned long long g_cliff_words[2];
namespace {
std::atomic<unsigned long long*> g_cliff_base[kMaxDevices];
}  // namespace

unsigned long long* cliff_counters()
{
    const int dev = current_device();
    if (dev < 0) return nullptr;
    unsigned long long* base = g_cliff_base[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_cliff_words)) != hipSuccess || sym == nullptr) {
            (void)hipGetLastError();
            return nullptr;
        }
        base = static_cast<unsigned long long*>(sym);
        g_cliff_base[dev].store(base, std::memory_order_release);
    }
    return base;
}

// sum over the devices this process has launched on; a blocking 16-byte copy per device (like fa_last_forward_route)
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

bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

uint32_t next_serial()
{
    uint32_t serial = g_serial.fetch_add(1, std::memory_order_relaxed);
    if (serial == 0) serial = g_serial.fetch_add(1, std::memory_order_relaxed);   // 0 is the ring's initial content
    return serial;
}

namespace {
// capture slots come back through a hipUserObject the capturing graph retains: its destructor runs when the graph and every executable
// instantiated from it are gone.  (No HIP call is allowed in there: it only pushes a number onto a list.)
struct CaptureSlotToken {
    int dev, slot;
};
void release_capture_slot(void* ptr)
{
    CaptureSlotToken* t = static_cast<CaptureSlotToken*>(ptr);
    if (t->dev >= 0 && t->dev < kMaxDevices) {
        std::lock_guard<std::mutex> g(g_slots[t->dev].mu);
        g_slots[t->dev].free_capture.push_back(t->slot);
        g_stats.capture_slots_recycled.fetch_add(1, std::memory_order_relaxed);
    }
    delete t;
}
// true: the graph being captured on `stream` now owns `slot` (it returns it when it dies)
bool tie_capture_slot_to_graph(hipStream_t stream, int dev, int slot)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, nullptr, nullptr) != hipSuccess || graph == nullptr) {
        (void)hipGetLastError();
        return false;
    }
    CaptureSlotToken* tok = new CaptureSlotToken{dev, slot};
    hipUserObject_t uo = nullptr;
    if (hipUserObjectCreate(&uo, tok, release_capture_slot, 1, hipUserObjectNoDestructorSync) != hipSuccess || uo == nullptr) {
        (void)hipGetLastError();
        delete tok;
        return false;
    }
    if (hipGraphRetainUserObject(graph, uo, 1, hipGraphUserObjectMove) != hipSuccess) {
        (void)hipGetLastError();
        tok->dev = -1;                       // the destructor then only frees the token
        (void)hipUserObjectRelease(uo, 1);
        return false;
    }
    return true;
}
}  // namespace

// The flag word of a chain that has no workspace (see above).  false = no slot to be had (or no device symbol): the caller then launches
// the always-correct kernel of the chain alone.  `hold` keeps the device's slot table locked until the chain is enqueued.
bool next_flag(FlagRef& f, hipStream_t stream, bool capturing, std::unique_lock<std::mutex>& hold)
{
    const int dev = current_device();
    if (dev < 0) return false;
    uint32_t* base = g_ring_base[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_flag_ring)) != hipSuccess || sym == nullptr) return false;
        base = static_cast<uint32_t*>(sym);
        void* sym2 = nullptr;
        if (hipGetSymbolAddress(&sym2, HIP_SYMBOL(g_stat_ring)) != hipSuccess || sym2 == nullptr) return false;
        g_stat_base[dev].store(static_cast<unsigned long long*>(sym2), std::memory_order_release);
        g_ring_base[dev].store(base, std::memory_order_release);
    }
    SlotTable& tb = g_slots[dev];
    hold = std::unique_lock<std::mutex>(tb.mu);
    int slot = -1;
    f.done = nullptr;
    if (capturing) {
        const int k = take_capture_slot(tb);
        if (k < 0) {
            hold.unlock();
            return false;
        }
        slot = kEagerSlots + k;
        hold.unlock();                                   // (the runtime may run a user-object destructor -- which takes this mutex -- inside its calls)
        (void)tie_capture_slot_to_graph(stream, dev, k);   // not tied: the slot is never handed out again, as in round 3
        hold.lock();
    } else {
        uint64_t key = reinterpret_cast<uint64_t>(stream);
        if (stream == hipStreamPerThread) {
            if (t_thread_number == 0) t_thread_number = g_thread_numbers.fetch_add(1, std::memory_order_relaxed);
            key = (1ull << 63) | t_thread_number;
        }
        slot = take_eager_slot(tb, key, [&](int i) {
            if (hipEventQuery(tb.eager[i].done) == hipSuccess) return true;
            (void)hipGetLastError();
            return false;
        });
        if (slot < 0) {
            hold.unlock();
            return false;
        }
        EagerSlot& e = tb.eager[slot];
        // A slot found BY KEY whose last chain has not completed: normally the same stream, ordered behind it anyway -- but a stream handle
        // can be recycled (hipStreamDestroy does not wait; the runtime may hand the same value to a new stream while the old stream's last
        // chain still runs), and two chains in flight would then share a word.  In the PRODUCT the word only reports (the fallback is
        // inside the kernel): a recycled handle can at worst misreport one route, and back-to-back forwards on one stream -- the normal
        // case, where the previous event is never complete yet -- pay nothing.  The ablation library's chains decide launches by the word:
        // there the new stream is ordered behind the slot's event (a no-op for the same stream).
#if FA_ABLATION
        if (e.state == 1 && e.done != nullptr && hipEventQuery(e.done) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamWaitEvent(stream, e.done, 0);
        }
#endif
        if (e.done == nullptr && hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            e.done = nullptr;   // without an event the slot can never change hands safely: it simply stays with its stream
        }
        f.done = e.done;
        f.dev = dev;
        f.eager_slot = slot;
    }
    f.word = base + slot;
    f.serial = next_serial();
    f.stats = g_stat_base[dev].load(std::memory_order_acquire) + 2 * (size_t)slot;
    return true;
}
// behind the chain's last launch, table still locked (`hold`): the event that tells when this slot may change hands
void chain_enqueued(const FlagRef& f, hipStream_t stream)
{
    if (f.dev < 0 || f.eager_slot < 0) return;
    EagerSlot& e = g_slots[f.dev].eager[f.eager_slot];
    if (f.done != nullptr && hipEventRecord(f.done, stream) == hipSuccess) {
        e.state = 1;
    } else {
        (void)hipGetLastError();
        e.state = 2;
    }
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
