// ldpc_mi355x.hip -- host side of libldpc_mi355x.so: the C ABI declared in
// include/ldpc_mi355x.h, Tanner-graph preparation, workspace management and
// kernel dispatch.  Device code lives in bp_kernels.hpp.
//
// Reference interfaces replaced (QuantumSavory/LDPCDecoders.jl):
//   BeliefPropagationDecoder(H, per, max_iters)  src/decoders/belief_propagation.jl:61-67
//   reset! + decode!                              :83-91, :121-188
//   batchdecode!                                  :220-231, src/decoders/abstract_decoder.jl:31-48
//
// There is deliberately no CPU path in this file: if HIP or the device is not
// usable every compute entry fails with a status code.
#include "../../include/ldpc_mi355x.h"
#include "../../include/ldpc_mi355x_debug.h"
#define LDPC_AUX_KERNELS 1
#include "bp_kernels.hpp"
#include "bp_lds_kernels.hpp"
#include "bp_node_kernels.hpp"
#include "bp_team_kernels.hpp"
#include "pickers.hpp"
#include "host_env.hpp"
#include "host_wait.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

using namespace ldpc;

namespace {

thread_local std::string g_err;

ldpc_status fail(ldpc_status st, const std::string &msg)
{
    g_err = msg;
    return st;
}

}  // namespace

namespace ldpc_detail {
// shared with osd_host.cpp: records the message ldpc_last_error() returns
ldpc_status set_error(ldpc_status st, const std::string &msg) { return fail(st, msg); }

// ---- bounded host-side waits (host_wait.hpp)
namespace {
std::atomic<int64_t> g_wait_limit_ms{600000};
constexpr int kMaxDev = 64;
std::atomic<bool> g_stalled[kMaxDev];
std::mutex g_stall_mu;
std::string g_stall_msg[kMaxDev];

ldpc_status expired(int device, const char *what, int64_t limit_ms)
{
    const std::string msg = std::string(what) + ": the device did not get there within " + std::to_string(limit_ms) +
                            " ms (ldpc_set_wait_limit_ms); device " + std::to_string(device) +
                            " is taken to be stalled: every later call on it fails with this message, and what it may still be "
                            "using is not freed";
    if (device >= 0 && device < kMaxDev) {
        std::lock_guard<std::mutex> lk(g_stall_mu);
        if (!g_stalled[device].load()) { g_stall_msg[device] = msg; g_stalled[device].store(true); }
    }
    return fail(LDPC_ERR_HIP, msg);
}

template <class Query>
ldpc_status poll_until(Query &&query, int device, const char *what)
{
    if (device_stalled(device)) return stalled_error(device);
    const int64_t limit = g_wait_limit_ms.load(std::memory_order_relaxed);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = query();
        if (q == hipSuccess) return LDPC_OK;
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            return fail(LDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
        }
        if (spins < 64) { __builtin_ia32_pause(); continue; }   // (a query is ~1 us: the first polls back to back)
        const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (limit > 0 && us > limit * 1000) return expired(device, what, limit);
        if (us < 300) __builtin_ia32_pause();                    // latency-bound calls (a small batch is ~100 us): keep polling
        else if (us < 5000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(us < 100000 ? 50 : 200));
    }
}
}  // namespace

int64_t wait_limit_ms() { return g_wait_limit_ms.load(std::memory_order_relaxed); }
bool device_stalled(int device) { return device >= 0 && device < kMaxDev && g_stalled[device].load(std::memory_order_acquire); }
ldpc_status stalled_error(int device)
{
    std::lock_guard<std::mutex> lk(g_stall_mu);
    return fail(LDPC_ERR_HIP, (device >= 0 && device < kMaxDev) ? g_stall_msg[device] : std::string("device stalled"));
}
ldpc_status wait_event(hipEvent_t e, int device, const char *what)
{
    if (wait_limit_ms() == 0 && !device_stalled(device)) {       // unbounded, as before round 4
        const hipError_t q = hipEventSynchronize(e);
        if (q == hipSuccess) return LDPC_OK;
        (void)hipGetLastError();
        return fail(LDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    }
    return poll_until([&] { return hipEventQuery(e); }, device, what);
}
ldpc_status wait_stream(hipStream_t s, int device, const char *what)
{
    if (wait_limit_ms() == 0 && !device_stalled(device)) {
        const hipError_t q = hipStreamSynchronize(s);
        if (q == hipSuccess) return LDPC_OK;
        (void)hipGetLastError();
        return fail(LDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    }
    return poll_until([&] { return hipStreamQuery(s); }, device, what);
}
// hipDeviceSynchronize has no query form: it runs in a helper thread that the caller waits for with the deadline; a
// thread that never comes back is left behind (detached) with the state it shares with nobody else.
ldpc_status wait_device(int device, const char *what)
{
    if (device_stalled(device)) return stalled_error(device);
    const int64_t limit = wait_limit_ms();
    if (limit == 0) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t q = hipSetDevice(device);
        if (q == hipSuccess) q = hipDeviceSynchronize();
        if (prev >= 0) (void)hipSetDevice(prev);
        if (q == hipSuccess) return LDPC_OK;
        (void)hipGetLastError();
        return fail(LDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(q));
    }
    struct Shared { std::mutex m; std::condition_variable cv; bool done = false; hipError_t e = hipSuccess; };
    auto sh = std::make_shared<Shared>();
    std::thread([sh, device] {
        hipError_t q = hipSetDevice(device);
        if (q == hipSuccess) q = hipDeviceSynchronize();
        if (q != hipSuccess) (void)hipGetLastError();
        std::lock_guard<std::mutex> lk(sh->m);
        sh->e = q; sh->done = true;
        sh->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(sh->m);
    if (!sh->cv.wait_for(lk, std::chrono::milliseconds(limit), [&] { return sh->done; })) {
        lk.unlock();
        return expired(device, what, limit);
    }
    if (sh->e == hipSuccess) return LDPC_OK;
    return fail(LDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(sh->e));
}
}  // namespace ldpc_detail

namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return fail(e_ == hipErrorOutOfMemory ? LDPC_ERR_OUT_OF_MEMORY : LDPC_ERR_HIP,     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
        }                                                                                      \
    } while (0)

// Current device of the calling thread, put back when the scope ends (pool eviction and the multi-device entries
// switch devices; the caller's choice must survive them).
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// A chunk group: physical chunks (hipMemCreate) mapped side by side -- each by ONE hipMemMap, access granted mapping by
// mapping -- at an aligned base inside a virtual reservation of its own.  This is how every message array of >= 1 GiB is
// backed (DESIGN.md "Workspace placement").  What it took to make ROCm 7.2's virtual-memory path dependable:
//   * Round 2: a kernel touching a freshly mapped group died of "Memory access fault ... Reason: Unknown" in about one
//     process out of six of tools/team_fault_hunt2.sh.  The faulting addresses (gpurun_out/s30, s33, s34 of round 2) are
//     base + 1 GiB + 0xfb3000 of a 2-chunk group (32 slots: row 33,921 of slot 31) and base + 2 GiB + 0xfb3000 of a
//     3-chunk group (64 slots: row 35,690 of slot 62), s35 base + 2 GiB + 0xe1f000: the SAME offset into the last chunk
//     for different rows, slots and group sizes -- a property of a page of the address range, not of the kernel's
//     indexing.  Not a race (a device synchronise and 3 ms of sleep after the mapping changed nothing).  What all
//     incidents shared: the reservation lay where hipMalloc'd buffers of the same process -- the script's earlier,
//     smaller workspaces and torch's tensors, whose sizes and hence addresses repeat from process to process -- had
//     lived and been freed.  That is what a deferred unmap of a freed range does when it is applied AFTER the range has
//     been mapped again: it clears the new mapping's entries from the old buffer's first page on.  A range that has
//     been mapped before is a RE-mapped range whoever mapped it first (tools/vmm_probe2.hip faulted the same way on its
//     own re-mappings).
//   * Hence the rule this code ENFORCES (round 3): a group is only ever mapped into an address window of its own
//     (16 ... 64 TiB, far below where hipMalloc and mmap hand out addresses), at addresses that a monotonic cursor
//     hands out ONCE per process.  A reservation that does not land where it was asked for is given back and the
//     buffer falls back to hipMalloc (big_alloc() -> LDPC_ERR_UNSUPPORTED); so does a request once the window is used
//     up.  No un-hinted reservation is ever mapped.  Since no address is handed out twice, a destroyed group's
//     reservation can be given back (hipMemAddressFree) without ever being mapped again: address space is bounded by
//     the live groups, and the cursor (48 TiB) by ~1,900 workspaces of the C3 size per process.
//   * Groups are POOLED per process: a buffer that is released hands its group, still mapped, to the pool, and the next
//     buffer of that size takes it over -- a process that keeps using the same few sizes creates each group once, and a
//     group that has been probed carries its grade, so a later workspace of that size skips the placement search.  A
//     group enters the pool only after its device has been synchronised (nothing enqueued may still use it).  The
//     pool holds at most 64 GiB; beyond that the oldest group is really unmapped and released; an allocation that
//     runs out of memory empties it, and so does ldpc_trim_memory().
constexpr uintptr_t kVmmWindowLo = (uintptr_t)16 << 40, kVmmWindowHi = (uintptr_t)64 << 40;

static bool vmm_log() { static const bool v = exp_env("LDPC_VMM_LOG") != nullptr; return v; }

struct ChunkGroup {
    void *resv = nullptr;
    size_t resv_size = 0;
    char *base = nullptr;
    size_t chunk = 0, mapped = 0;            // chunk size; chunks mapped so far
    std::vector<hipMemGenericAllocationHandle_t> h;
    int device = 0;
    float probe_tbs = 0.f;                   // what the placement probe measured on it as a whole workspace (0 = never probed)
    size_t probe_bytes = 0;                  // ... of this many bytes
    size_t bytes() const { return chunk * mapped; }
    bool empty() const { return resv == nullptr; }
    void destroy()                           // really give the memory and the address range back (device idle: see above)
    {
        if (!resv) return;
        DeviceGuard guard;
        (void)hipSetDevice(device);
        // (bounded, host_wait.hpp: a device that does not drain keeps its mappings -- leaked, never unmapped under a kernel)
        if (ldpc_detail::device_idle_for_release(device, "chunk group release (device synchronise before the unmap)")) {
            if (vmm_log()) std::fprintf(stderr, "[ldpc-vmm] unmap %p .. %p\n", (void *)base, (void *)(base + bytes()));
            for (size_t k = 0; k < mapped; ++k) (void)hipMemUnmap(base + k * chunk, chunk);   // one unmap per map
            for (auto q : h) (void)hipMemRelease(q);
            if (ldpc_detail::device_idle_for_release(device, "chunk group release (device synchronise after the unmap)"))
                (void)hipMemAddressFree(resv, resv_size);   // never handed out again: the cursor only moves up
            (void)hipGetLastError();
        }
        h.clear();
        resv = nullptr; base = nullptr;
        resv_size = chunk = mapped = 0;
    }
};

namespace {
std::mutex g_pool_mu;
std::vector<ChunkGroup> g_pool;   // oldest first

size_t pool_cap_bytes()
{
    static const size_t v = [] { const char *e = exp_env("LDPC_POOL_GIB"); return (size_t)(e ? std::max(0, std::atoi(e)) : 64) << 30; }();
    return v;
}
void pool_put(ChunkGroup &&g)
{
    if (g.empty()) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool.push_back(std::move(g));
    size_t total = 0;
    for (const ChunkGroup &q : g_pool) total += q.bytes();
    while (total > pool_cap_bytes() && !g_pool.empty()) {
        total -= g_pool.front().bytes();
        g_pool.front().destroy();
        g_pool.erase(g_pool.begin());
    }
}
// the pooled group of this chunk size on this device that fits `bytes` most tightly (ties: the better probed one)
bool pool_take(size_t bytes, size_t chunk, int device, ChunkGroup *out)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int best = -1;
    for (int k = 0; k < (int)g_pool.size(); ++k) {
        const ChunkGroup &q = g_pool[(size_t)k];
        if (q.device != device || q.chunk != chunk || q.bytes() < bytes) continue;
        if (best < 0 || q.bytes() < g_pool[(size_t)best].bytes() ||
            (q.bytes() == g_pool[(size_t)best].bytes() && q.probe_tbs > g_pool[(size_t)best].probe_tbs))
            best = k;
    }
    if (best < 0) return false;
    *out = std::move(g_pool[(size_t)best]);
    g_pool.erase(g_pool.begin() + best);
    return true;
}
void pool_drop_all()
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (ChunkGroup &q : g_pool) q.destroy();
    g_pool.clear();
}

// The next address of the private window for a reservation of `bytes` (a multiple of `align`), or 0 when the window is
// used up (or switched off: LDPC_VMM_HINT_TIB=0 in the experiments build) -- the caller then does without a chunk group.
uintptr_t vmm_next_address(size_t bytes, size_t align)
{
    static const uintptr_t lo = [] { const char *e = exp_env("LDPC_VMM_HINT_TIB"); return e ? (uintptr_t)std::atoll(e) << 40 : kVmmWindowLo; }();
    static std::atomic<uintptr_t> cursor{lo};
    if (lo == 0) return 0;
    const uintptr_t step = ((uintptr_t)bytes + 2 * align + align - 1) / align * align;   // (a gap between neighbours)
    const uintptr_t at = cursor.fetch_add(step);
    if (at < lo || at + step > std::max(kVmmWindowHi, lo + ((uintptr_t)1 << 40))) return 0;
    return at;
}
}  // namespace

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int device = -1;  // where p lives (set when it is allocated)
    ChunkGroup grp;   // set when the buffer is a chunk group (ensure_chunked), empty for hipMalloc
    ldpc_status ensure(size_t bytes)
    {
        if (bytes <= cap) return LDPC_OK;
        release();
        if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); device = -1; }
        if (ldpc_detail::device_stalled(device)) return ldpc_detail::stalled_error(device);
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipErrorOutOfMemory) {   // the pool may be what is in the way
            (void)hipGetLastError();
            pool_drop_all();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return fail(LDPC_ERR_OUT_OF_MEMORY, "hipMalloc(" + std::to_string(bytes) + " B): " + hipGetErrorString(e));
        }
        cap = bytes;
        return LDPC_OK;
    }
    // A chunk group of at least `bytes`: out of the pool (unless `fresh`: the placement search wants OTHER physical
    // memory than what it has seen) or newly created -- chunks of `chunk` bytes, base aligned to `align`; `shuffle` maps
    // them in a scrambled order (experiments).  LDPC_ERR_UNSUPPORTED: no chunk group to be had under the rules above
    // (the caller uses hipMalloc).
    ldpc_status ensure_chunked(size_t bytes, size_t chunk, size_t align, bool shuffle, int device, bool fresh = false)
    {
        if (bytes <= cap) return LDPC_OK;
        release();
        if (!fresh && !shuffle && pool_take(bytes, chunk, device, &grp)) {
            p = grp.base;
            cap = grp.bytes();
            return LDPC_OK;
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            bool refused = false;
            const hipError_t e = create(bytes, chunk, align, shuffle, device, &refused);
            if (e == hipSuccess) return LDPC_OK;
            (void)hipGetLastError();
            grp.destroy();
            p = nullptr; cap = 0;
            if (refused) return fail(LDPC_ERR_UNSUPPORTED, "chunk group: no address range of the private window to be had");
            if (e != hipErrorOutOfMemory || attempt == 1)
                return fail(e == hipErrorOutOfMemory ? LDPC_ERR_OUT_OF_MEMORY : LDPC_ERR_HIP, std::string("chunk group: ") + hipGetErrorString(e));
            pool_drop_all();   // out of memory: what the pool holds may be what is missing
        }
        return LDPC_ERR_OUT_OF_MEMORY;
    }
    hipError_t create(size_t bytes, size_t chunk, size_t align, bool shuffle, int device, bool *refused)
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        const size_t n = (bytes + chunk - 1) / chunk;
        grp = ChunkGroup();
        grp.device = device;
        // An address of the private window, used once; a reservation that lands anywhere else is given back unmapped.
        const size_t want = n * chunk + align;
        const uintptr_t hint = vmm_next_address(want, align);
        if (!hint) { *refused = true; return hipErrorInvalidValue; }
        void *got = nullptr;
        hipError_t e = hipMemAddressReserve(&got, want, 0, (void *)hint, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); *refused = true; return e; }
        if (vmm_log()) std::fprintf(stderr, "[ldpc-vmm] reserve hint %p -> %p\n", (void *)hint, got);
        if ((uintptr_t)got != hint) {
            (void)hipMemAddressFree(got, want);
            (void)hipGetLastError();
            *refused = true;
            return hipErrorInvalidValue;
        }
        grp.resv = got;
        grp.resv_size = want;
        grp.chunk = chunk;
        grp.base = (char *)(((uintptr_t)grp.resv + align - 1) / align * align);
        for (size_t k = 0; k < n; ++k) {
            hipMemGenericAllocationHandle_t q;
            if ((e = hipMemCreate(&q, chunk, &prop, 0)) != hipSuccess) return e;
            grp.h.push_back(q);
        }
        std::vector<size_t> order(n);
        for (size_t k = 0; k < n; ++k) order[k] = k;
        if (shuffle) {
            unsigned long long sd = 88172645463325252ull;
            for (size_t i = n - 1; i > 0; --i) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(order[i], order[sd % (i + 1)]); }
        }
        for (size_t k = 0; k < n; ++k) {
            if ((e = hipMemMap(grp.base + k * chunk, chunk, 0, grp.h[order[k]], 0)) != hipSuccess) return e;
            grp.mapped = k + 1;
            if ((e = hipMemSetAccess(grp.base + k * chunk, chunk, &acc, 1)) != hipSuccess) return e;
        }
        p = grp.base;
        cap = n * chunk;
        if (vmm_log()) std::fprintf(stderr, "[ldpc-vmm] map   %p .. %p (%zu chunks)\n", (void *)grp.base, (void *)(grp.base + cap), n);
        return hipSuccess;
    }
    void swap(DevBuf &o)
    {
        std::swap(p, o.p);
        std::swap(cap, o.cap);
        std::swap(device, o.device);
        std::swap(grp, o.grp);
    }
    // device_idle: the caller has just waited for the device (ldpc_bp_destroy).  Every wait here is bounded
    // (host_wait.hpp): memory of a device that does not drain is leaked, not freed under a kernel that may still run.
    void release(bool device_idle = false)
    {
        if (!grp.empty()) {
            // hipFree waits for the device by itself; a group that goes to the pool still mapped must do so explicitly:
            // kernels of earlier asynchronous calls may still be using it when the next owner takes it over
            if (device_idle || ldpc_detail::device_idle_for_release(grp.device, "workspace release (device synchronise before pooling)"))
                pool_put(std::move(grp));
            else { grp.h.clear(); grp.resv = nullptr; }   // (leaked)
        } else if (p) {
            if (device_idle || device < 0 || ldpc_detail::device_idle_for_release(device, "hipFree (device synchronise before the free)"))
                (void)hipFree(p);
        }
        grp = ChunkGroup();
        p = nullptr;
        cap = 0;
    }
};

// How many syndromes per workgroup pass the LDS kernel holds (log2), or -1 if even one
// syndrome's messages do not fit.  Workgroups per CU come first (the check sweep is VALU-bound, the
// other phases are latency-bound: co-resident workgroups in different phases fill each other's
// gaps), then the largest S that still admits that many: 3 workgroups of 8 waves is what the
// register budget allows (2 when LLRs are written).  Measured on n=1008: S=1 x 3 workgroups 60.0 M
// syndromes/s, S=2 x 2 51.6 M, S=4 x 1 31.6 M; BB-72: S=16 x 3 best.
int lds_logS(int64_t s, int64_t n, int64_t nnz, bool want_llr)
{
    constexpr size_t kLds = 160 * 1024;
    if (nnz >= 65535 || s >= 65535 || n >= 65535) return -1;   // uint16 graph copies in LDS
    for (int wgs = want_llr ? 2 : 3; wgs >= 1; --wgs) {
        int best = -1;
        for (int l = 0; l <= 6; ++l)
            if ((lds_bytes_needed((int)s, (int)n, (int)nnz, 1 << l, want_llr) + 512) * (size_t)wgs <= kLds) best = l;
        if (best >= 0) return best;
    }
    return -1;
}

}  // namespace

struct ldpc_bp_decoder {
    int64_t s = 0, n = 0, nnz = 0, max_iters = 0;
    double per = 0.0;
    int device = 0;
    int num_cus = 0;
    int max_cdeg = 0, max_bdeg = 0;
    int wpt_fixed = 0;        // waves per tile requested by the caller, 0 = chosen per batch
    int resident_fixed = 0;   // workspace slots requested by the caller, 0 = fill the chip
    int last_threads = 512, last_grid = 0;   // geometry of the most recent streaming launch (info)
    int last_kernel = 0, last_team = 1;      // which kernel the most recent call ran (info)
    int last_lds_rows = 0;                   // ... and how many message rows each team member kept in LDS
    int last_rows_on_chip = 0;               // ... and how many rows of a tile lived in LDS and registers in all
    int rows_on_chip = 0;                    // (of the tables in hand: team_rows_build())
    size_t ws_budget = 0;     // bytes the message workspace may take
    int blocks_cache[2][17];  // [want_llr][waves per tile] -> resident workgroups per CU, -1 = not queried yet
    int variant = 0;          // 0 auto, 1 HBM-streaming tile kernel, 2 LDS-resident kernel, 3 node-parallel kernel
    bool node_ok = false;     // syndrome + decision bytes of one syndrome fit the LDS (bp_node_kernels.hpp)
    bool node_msg_lds = false;   // ... and its nnz messages too (LDPC_NODE_MSG_LDS=0 keeps them in the global slots)
    int node_split_check = 0, node_split_edge = 0;   // hybrid placement: checks [0, split) keep their messages in LDS (0 = off)
    int64_t node_max_batch = 0;   // auto: largest batch the node-parallel kernel takes where the team kernel does not apply
    int64_t node_take_max = 0;    // most stragglers the node-parallel kernel takes as the second pass of the hand-off
    DevBuf node_msg;          // [workgroups][nnz] double, the node-parallel kernel's message slots
    // team kernel (bp_team_kernels.hpp): arrival counters + mismatch words, and the host-mapped fault word
    DevBuf team_ws, team_ws_lvl[2];   // control blocks of the team kernel: level 0 / the passes over the packed levels
    // rows in LDS (bp_team_kernels.hpp, TeamRows): regular graphs of the (8,4) bucket keep the host copy of csc2csr;
    // the tables are built for the member count of the first persistent launch (team_rows_build())
    std::vector<int> h_csc2csr;
    // irregular graphs (and regular ones without a rows-on-chip instantiation): host copies of the whole graph for
    // team_irr_tables() -- whole checks in the LDS of their owners (bp_team_kernels.hpp, IRR)
    std::vector<int> h_row_ptr, h_edge_bit, h_col_ptr, h_irr_c2r;
    int irr_G = 0, irr_R = 0, irr_on_chip = 0;   // what the tables below were built for: members, LDS rows per member, rows in LDS in all
    int irr_dcb = 0;                             // check-degree bucket of the IRR instantiation for this graph (team_irr_dc_bucket(); 0: none)
    DevBuf irr_ctab, irr_ptab, irr_ploc, irr_lds_edge, irr_posmap;
    bool team_rows_on = true;         // LDPC_TEAM_ROWS at create (0 = every row in the slot)
    int rows_dc = 0, rows_dv = 0;     // the graph's (check, bit) degree when it is regular and has a rows-in-LDS instantiation
    int rows_G = 0, rows_R = 0;       // what the tables below were built for: members per team, LDS rows per member
    int rows_first_c = 0;             // ... leading chunks of every member whose checks are whole on chip (TeamRows::first_c)
    int rows_regs = 0, rows_static_c = 0, rows_static_v = 0;   // ... register rows per wave, chunks per sweep that waves own by right
    DevBuf rows_ctab, rows_vtab, rows_lds_edge, rows_reg_edge, rows_posmap;   // (posmap [n]: position of every bit in the dealt order, for unpack_llr_kernel)
    int team_regs = kTeamRegRows;     // LDPC_TEAM_REGS: rows a wave may keep in its top registers (0 = none)
    bool team_pre_set = false;
    int team_pre = 2;                 // LDPC_TEAM_PRE: chunks of on-chip checks a wave updates between arriving at the barrier after the variable sweep and waiting at it (TeamParams::pre)
    int team_flip = 3;                // LDPC_TEAM_FLIP: bit 0 / 1: upper half of the waves walks its check / position chunks by right backwards (TeamRows::flip)
    int team_static_quarters = 3;     // LDPC_TEAM_STATIC: quarters of a member's chunks per sweep that its waves own by right (0: only each wave's first)
    unsigned int *team_fault = nullptr, *team_fault_dev = nullptr;
    // (LDPC_TEAM_PER_CU, LDPC_TEAM_MIN_ROWS, LDPC_TEAM_NO_MARGIN, LDPC_TEAM_ALWAYS_RELEASE, LDPC_TEAM_COOP_LAUNCH: read at create
    //  like the other knobs, so that a test or the fuzzer can vary them from decoder to decoder)
    int team_per_cu_want = 2;       // team workgroups per CU the plan asks for
    int64_t team_min_rows = 2048;   // message rows per sweep a member must have ...
    bool team_min_rows_set = false; // ... given by the environment (then also for the one team of an XCD: kTeamMinRowsOne does not apply)
    bool team_no_margin = false, team_always_release = false, team_coop = false;
    int team_max = 32;        // workgroups per tile at most (LDPC_TEAM_MAX; 1 = team kernel off)
    bool team_max_set = false;   // ... given by the environment: no automatic large teams for batches of <= 3 tiles
    // (read at create, so that a test or the fuzzer can vary them from decoder to decoder)
    size_t team_cache = (size_t)240 << 20;   // LDPC_TEAM_CACHE_MIB: message slots in flight that persistent teams may hold (team_fit())
    int team_xcds = 0;        // LDPC_TEAM_XCDS: XCDs that host persistent teams, fixed (experiments; 0 = team_fit() chooses)
    int team_dynamic = 1;     // LDPC_TEAM_DYNAMIC: a member's waves take its chunks from a counter in LDS (0: every W-th)
    int team_pairs = 3;       // LDPC_TEAM_PAIRS: bit 0: two nodes of the full degree are loaded together; bit 1: the four bits of a position chunk (rows-on-chip kernels)
    int team_ahead_from = 1;  // LDPC_TEAM_AHEAD_FROM: first iteration whose test may ride with the next check sweep (1: realistic -1.6 %, waterfall -0.7 % against 2, profiles/r03_ahead_from1.txt)
    bool team_ahead_set = false;   // LDPC_TEAM_AHEAD given (else: 1 for a single round of teams over all XCDs)
    bool llr_exact = false;   // ldpc_bp_options.llr_exact: LLRs from the full posterior odds (bp_kernels.hpp llr_of)
    int team_llr_raw = 4;     // LDPC_TEAM_LLR_RAW (4 becomes 5 with llr_exact): what the team kernel's variable sweep stores per bit when LLRs are wanted (TeamParams::llr_raw)
    bool team_llr_footprint = true;   // LDPC_TEAM_LLR_FOOTPRINT: the plan counts a tile's LLR rows (n x 512 B, rewritten in every iteration) as part of its slot
    int team_ahead = 32;      // LDPC_TEAM_AHEAD: active lanes from which on a team starts the next check sweep with the convergence
                              // test still under way (two team barriers an iteration instead of three); 0 = never
    // latency mode of the host-pointer entry (tiny batches, a plain decode!): the kernel reads and writes a
    // host-mapped staging image and raises a flag in it; no copies, no events, no stream synchronisation
    DevBuf done_ctr;          // one device word, zero between launches
    void *lat_pin = nullptr;  // [flag 256 B][syndromes][errors][converged][iters][llr], hipHostMallocMapped
    void *lat_pin_dev = nullptr;
    size_t lat_pin_cap = 0;
    unsigned lat_ticket = 0;
    // kernels whose dynamic-LDS limit is set / whose occupancy is known (both are ~us runtime calls)
    std::vector<std::pair<const void *, int>> kernel_info;
    ldpc_status prepare_kernel(const void *fn, int threads, size_t lds, int *per_cu);
    int lds_logS[2] = {-1, -1};   // [want_llr]: syndromes per workgroup pass of the LDS kernel, -1 = does not fit
    // device graph
    DevBuf row_ptr, edge_bit, col_ptr, csc2csr;
    // workspace
    DevBuf msg;               // [resident_tiles][nnz][64] double
    DevBuf ctrl;              // queue (u32) + sum_iters (u64), 64 B
    DevBuf cold;              // BPCold blocks of the passes of the call in flight (bp_kernels.hpp)
    // per-batch buffers (grow only)
    DevBuf synmask, nevermask, errmask, finmask, llr_t;
    // packed levels of the straggler hand-off (decode_device_impl): message tiles, batch positions, iteration
    // counts, and the per-level images of the per-batch buffers above
    DevBuf lvl_state[2], lvl_list[2], lvl_it[2], lvl_syn[2], lvl_never[2], lvl_err[2], lvl_fin[2], lvl_llr[2];
    int defer_thresh = 0;     // 0 auto (16 lanes), -1 off, else the lane count at which a tile gives up
    int defer_t0 = 16, defer_t1 = 16;   // lanes at which a fresh tile / a packed tile of level 1 hands off (LDPC_DEFER_T0 / _T1 at create; T1 0 = one level)
    int lvl_cap_force = 0;    // tests (LDPC_DEFER_CAP_TILES at create): packed tiles per level, so that full levels are exercised
    float placement_ms = 0.f; // probe time of the chosen workspace allocation (0 = no probing happened)
    int placement_candidates = 0;
    // staging for the host-pointer entry
    DevBuf st_all;            // device image of a small host batch
    void *pin = nullptr;      // pinned host image for small batches
    size_t pin_cap = 0;
    // host-buffer pipeline for large batches: 3 pinned + 3 device chunk images, copy / compute / copy streams
    static constexpr int kPipe = 3;
    void *pipe_pin[kPipe] = {};
    size_t pipe_pin_cap = 0;
    DevBuf pipe_dev[kPipe];
    hipStream_t pipe_stream[3] = {};          // H2D, compute, D2H
    hipEvent_t pipe_ev[kPipe][3] = {};        // per slot: H2D done, compute done, D2H done
    // timing ring: the last kRing batch calls keep their HIP events and iteration sums
    // (the API reaches 16 calls back; the ring is twice that so that a kernel may zero the control
    // slot of the call after it -- which is the slot of the call 31 back -- without touching history)
    static constexpr int kRing = 32;
    static constexpr int kHistory = 16;
    hipEvent_t ev[kRing][4] = {};
    bool timed[kRing] = {};
    bool two_events[kRing] = {};   // the call recorded ev[1] / ev[2] only (single-kernel paths)
    bool ctrl_clean[kRing] = {};   // the slot's 64 control bytes are known to be zero
    uint64_t ncalls = 0;
    // Calls on one handle share its workspace, so they execute in call order whatever streams they are
    // given: a call that arrives on another stream than its predecessor first waits for that one's last event.
    hipStream_t last_stream = nullptr;
    hipEvent_t last_ev = nullptr;  // one of ev[][] (not owned): completion of the most recent enqueued call
    int inject_fault = 0;          // tests (experiments build, LDPC_TEAM_INJECT_FAULT at create): 1 = team kernels raise the fault word at once, 2 = a member misses the roll call
    unsigned rollcall_ticks = 2000000u;   // 20 ms of the 100 MHz clock: how long the members of a team wait for each other at launch (team_rollcall)

    bool device_idle = false;      // ldpc_bp_destroy has waited for the device: frees need not wait again
    ~ldpc_bp_decoder()
    {
        const bool stalled = ldpc_detail::device_stalled(device);   // (then nothing is freed: host_wait.hpp)
        const bool idle = device_idle && !stalled;
        DevBuf *all[] = {&row_ptr, &edge_bit, &col_ptr, &csc2csr, &msg, &ctrl, &synmask, &nevermask,
                         &errmask, &finmask, &llr_t, &st_all, &node_msg, &done_ctr, &team_ws, &team_ws_lvl[0], &team_ws_lvl[1], &cold,
                         &rows_ctab, &rows_vtab, &rows_lds_edge, &rows_reg_edge, &rows_posmap,
                         &irr_ctab, &irr_ptab, &irr_ploc, &irr_lds_edge, &irr_posmap};
        for (DevBuf *b : all) b->release(idle);
        for (int l = 0; l < 2; ++l)
            for (DevBuf *b : {&lvl_state[l], &lvl_list[l], &lvl_it[l], &lvl_syn[l], &lvl_never[l], &lvl_err[l], &lvl_fin[l], &lvl_llr[l]}) b->release(idle);
        for (DevBuf &b : pipe_dev) b.release(idle);
        if (stalled) return;   // pinned memory the device may still write, streams it may still run: left alone
        if (pin) (void)hipHostFree(pin);
        if (lat_pin) (void)hipHostFree(lat_pin);
        if (team_fault) (void)hipHostFree(team_fault);
        for (void *&q : pipe_pin)
            if (q) (void)hipHostFree(q);
        for (hipStream_t &q : pipe_stream)
            if (q) (void)hipStreamDestroy(q);
        for (auto &row : pipe_ev)
            for (hipEvent_t &e : row)
                if (e) (void)hipEventDestroy(e);
        for (auto &slot : ev)
            for (hipEvent_t &e : slot)
                if (e) (void)hipEventDestroy(e);
    }
};

ldpc_status ldpc_bp_decoder::prepare_kernel(const void *fn, int threads, size_t lds, int *per_cu)
{
    for (auto &ki : kernel_info)
        if (ki.first == fn) { *per_cu = ki.second; return LDPC_OK; }
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        nb = 1;
    }
    kernel_info.emplace_back(fn, nb);
    *per_cu = nb;
    return LDPC_OK;
}

namespace {

// memcpy spread over a few host threads: one thread moves ~10 GB/s, PCIe Gen5 x16 wants ~50
void parallel_memcpy(void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::max<unsigned>(hw / 2, 1), 8);
    nt = std::min(nt, bytes / ((size_t)4 << 20));
    if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = ((bytes + nt - 1) / nt + 4095) & ~(size_t)4095;
    for (size_t t = 0; t < nt; ++t) {
        const size_t lo = t * per, hi = std::min(bytes, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=] { std::memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    for (auto &t : th) t.join();
}

}  // namespace

// bytes added to every workspace slot (LDPC_SLOT_PAD at build time; LDPC_SLOT_PAD_BYTES overrides it
// at run time for experiments; kept a multiple of 512 so that rows stay aligned)
static size_t slot_pad_bytes()
{
    static const size_t v = [] {
        size_t p = LDPC_SLOT_PAD;
        if (const char *e = exp_env("LDPC_SLOT_PAD_BYTES")) p = (size_t)std::atoll(e);
        return p & ~(size_t)511;
    }();
    return v;
}

// Backing of the big message arrays (the workspace of the tile / team kernels, the packed tiles of the hand-off
// levels).  C3 full-50, same binary, one process each (tools/ws_alloc_matrix.sh): a plain hipMalloc of the 24.75 GiB
// workspace runs 1.19 s or 1.42 s per launch depending on what the allocator hands out (the "placement classes" of
// round 1); the same bytes as 1 GiB hipMemCreate chunks mapped at a 1 GiB-aligned virtual base 1.163 s in every
// run (shuffled chunk order the same; 64 MiB chunks 1.170-1.179 s, 256 MiB 1.203 s, one 32 GiB handle 1.180 s).
// What differs is the translation, not the memory: the driver can only use large page-table fragments where the
// virtual and the physical address are aligned alike, and the sweeps keep ~12,000 distinct 2 MiB pages hot (768
// slots x 16 rows in flight).  Returns LDPC_ERR_UNSUPPORTED when the request is small (or LDPC_WS_ALLOC=malloc):
// the caller then uses hipMalloc.
static ldpc_status big_alloc(DevBuf &b, size_t bytes, int device, bool fresh = false)
{
    static const std::string ws_alloc = [] { const char *e = exp_env("LDPC_WS_ALLOC"); return std::string(e ? e : ""); }();
    if (ws_alloc == "malloc" || bytes < ((size_t)1 << 30)) return LDPC_ERR_UNSUPPORTED;
    size_t chunk = (size_t)1 << 30;
    bool shuffle = false;
    if (ws_alloc.rfind("vmm:", 0) == 0) {
        chunk = (size_t)std::max(2, std::atoi(ws_alloc.c_str() + 4)) << 20;
        shuffle = ws_alloc.find(":shuffle") != std::string::npos;
    }
    if (bytes <= b.cap) return LDPC_OK;
    const ldpc_status st = b.ensure_chunked(bytes, chunk, std::max<size_t>(chunk, (size_t)1 << 30), shuffle, device, fresh);
    if (st == LDPC_OK || st == LDPC_ERR_OUT_OF_MEMORY) return st;
    return LDPC_ERR_UNSUPPORTED;   // the virtual-memory API failed for another reason: fall back to hipMalloc
}

// Workspace placement.  What the sweeps can stream from a C3-size workspace (768 slots x 33 MiB) depends on the
// PHYSICAL memory behind it, in classes: ~6.0 / ~5.7 / ~5.1 TB/s on the variable sweep's pattern, launches of
// 1.16 / 1.20 / 1.42 s.  Round 2's probes (tools/vmm_probe*.hip, profiles/README.md) narrowed it down:
//   * not the virtual address (one physical allocation mapped at 40 bases: identical speed), as long as the base is
//     aligned like the physical chunks (1 GiB): a base that is only 2 / 4 MiB-aligned costs 3-8 %;
//   * a property of the physical chunks that stays with them wherever and in whatever order they are mapped, shared
//     by chunks allocated one after the other (tools/vmm_probe7.hip: group A 5.07, group B 5.72 TB/s; the 25 best
//     of the 50 = B, the 25 worst = A, each as fast as its group), the same in every slot of the workspace
//     (tools/vmm_probe6.hip) -- so a search has to look at groups of chunks, not at single ones;
//   * plain hipMalloc gets the slow class most of the time in a fresh process (5 of 6), 1 GiB chunks at a 1 GiB-
//     aligned base the fast one most of the time (22 of 26 bench runs on three boxes; never on a fourth).
// So: the workspace is a group of 1 GiB chunks (out of the pool if a graded one of that size is there); it is probed
// with both sweeps' patterns (one untimed first-touch pass, one timed pass, ~35 ms); unless it is of the fast class in
// absolute terms another group is allocated WHILE the first is held (freed memory would come straight back) and
// probed, and the better one kept; up to three more (LDPC_PLACEMENT_ROUNDS, default 5; 1 = take the first) only
// while the best is still below 5.75 TB/s, and never beyond half of the free HBM.  The losers are unmapped and
// released at once.  Transient HBM: 1x when the first group is fast, usually 2x, at most rounds x.  First-call cost
// ~0.1 s per group.
static ldpc_status ensure_workspace(ldpc_bp_decoder *d, size_t bytes, int grid, size_t slot_stride_bytes,
                                    hipStream_t stream)
{
    if (bytes <= d->msg.cap) return LDPC_OK;
    d->msg.release();
    const bool verbose = exp_env("LDPC_PLACEMENT_VERBOSE") != nullptr;
    static const int max_rounds = [] { const char *e = exp_env("LDPC_PLACEMENT_ROUNDS"); return e ? std::max(1, std::min(8, std::atoi(e))) : 5; }();
    d->placement_ms = 0.f;
    d->placement_candidates = 0;
    DevBuf first;
    ldpc_status st = big_alloc(first, bytes, d->device);
    if (st == LDPC_ERR_UNSUPPORTED) return d->msg.ensure(bytes);   // small, or LDPC_WS_ALLOC=malloc
    if (st != LDPC_OK) return st;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (max_rounds < 2 || d->nnz < 4 || grid < 64 || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) {
        (void)hipGetLastError();
        if (ea) (void)hipEventDestroy(ea);
        d->msg.swap(first);
        return LDPC_OK;
    }
    const long long stride = (long long)(slot_stride_bytes / sizeof(double));
    const double probe_bytes = 4.0 * (double)grid * (double)d->nnz * 512.0;   // both sweeps' patterns, read + write
    auto probe = [&](void *q) -> float {   // ms of one timed pass (after an untimed first-touch pass); <0 on error
        hipLaunchKernelGGL(placement_probe_kernel, dim3((unsigned)grid), dim3(512), 0, stream, (double *)q, stride, (int)d->nnz);
        if (hipEventRecord(ea, stream) != hipSuccess) return -1.f;
        hipLaunchKernelGGL(placement_probe_kernel, dim3((unsigned)grid), dim3(512), 0, stream, (double *)q, stride, (int)d->nnz);
        float ms = -1.f;
        if (hipEventRecord(eb, stream) != hipSuccess || ldpc_detail::wait_event(eb, d->device, "workspace placement probe") != LDPC_OK ||
            hipEventElapsedTime(&ms, ea, eb) != hipSuccess)
            return -1.f;
        return ms;
    };
    // the fast class in absolute terms: >= 6.0 TB/s of probe traffic (what 512+ streaming workgroups reach; a smaller
    // grid -- the team kernel's medium batches -- cannot be judged that way and takes the first group)
    // a group out of the pool that has been probed as a workspace of this size before keeps its grade
    auto graded = [&](const DevBuf &b) { return b.grp.probe_bytes == bytes && b.grp.probe_tbs > 0.f; };
    auto tbs_of = [&](float t) { return t > 0 ? (float)(probe_bytes / ((double)t * 1e-3) / 1e12) : 0.f; };
    std::vector<DevBuf> held;
    held.emplace_back();
    held.back().swap(first);
    std::vector<float> tbs;   // TB/s of probe traffic per held group
    if (graded(held[0])) tbs.push_back(held[0].grp.probe_tbs);
    else {
        tbs.push_back(tbs_of(probe(held[0].p)));
        held[0].grp.probe_tbs = tbs[0]; held[0].grp.probe_bytes = bytes;
    }
    if (verbose) std::fprintf(stderr, "[ldpc] workspace group 0 @%p: %.2f TB/s\n", held[0].p, tbs[0]);
    size_t best = 0;
    int probed = 1;
    if (grid >= 512) {
        // How far to look: a group of the fast class in absolute terms (>= 6.0 TB/s) ends the search; otherwise a second
        // group is always tried and the better of the two kept (the probe's scale moves by a few per cent from box to box
        // -- a box whose groups all probed 5.45-5.82 ran the kernel at 1.167 s on the 5.82 one -- so the comparison
        // inside a box is worth more than the absolute number); a third, fourth and fifth only while the best is still
        // below 5.75 TB/s (a 5.37 group ran the kernel at 1.35 s, the 5.0 class at 1.42 s).
        for (int r = 1; r < max_rounds && tbs[best] < 6.0f && !(r >= 2 && tbs[best] >= 5.75f); ++r) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
            if (bytes > free_b / 2) break;
            DevBuf cand;
            if (big_alloc(cand, bytes, d->device, /*fresh=*/true) != LDPC_OK) break;   // OTHER physical memory than the pool's
            held.emplace_back();
            held.back().swap(cand);
            tbs.push_back(tbs_of(probe(held.back().p)));
            held.back().grp.probe_tbs = tbs.back(); held.back().grp.probe_bytes = bytes;
            ++probed;
            if (verbose) std::fprintf(stderr, "[ldpc] workspace group %d @%p: %.2f TB/s\n", r, held.back().p, tbs.back());
            if (tbs.back() > tbs[best]) best = held.size() - 1;
        }
    }
    const ldpc_status wst = ldpc_detail::wait_stream(stream, d->device, "workspace placement probe (stream synchronise)");
    (void)hipGetLastError();
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    if (wst != LDPC_OK) return wst;   // (the groups held are leaked or pooled by their destructors' own bounded waits)
    d->msg.swap(held[best]);
    d->placement_ms = tbs[best] > 0 ? (float)(probe_bytes / ((double)tbs[best] * 1e12) * 1e3) : 0.f;
    d->placement_candidates = probed;
    for (DevBuf &b : held) {              // the groups that lost are given back for real: known to be no better,
        if (!b.grp.empty()) b.grp.destroy();   // they would only sit in the pool
        b.p = nullptr; b.cap = 0;
    }
    if (verbose) std::fprintf(stderr, "[ldpc] workspace: kept group %zu (%.2f TB/s) out of %zu held\n", best, tbs[best], tbs.size());
    return LDPC_OK;
}

extern "C" {

int32_t ldpc_abi_version(void) { return LDPC_MI355X_ABI_VERSION; }

ldpc_status ldpc_set_wait_limit_ms(int64_t ms)
{
    if (ms < 0) return fail(LDPC_ERR_INVALID_ARGUMENT, "negative wait limit");
    ldpc_detail::g_wait_limit_ms.store(ms);
    return LDPC_OK;
}
int64_t ldpc_get_wait_limit_ms(void) { return ldpc_detail::wait_limit_ms(); }

ldpc_status ldpc_trim_memory(void)
{
    pool_drop_all();
    return LDPC_OK;
}
const char *ldpc_build_target(void) { return "gfx950"; }
const char *ldpc_last_error(void) { return g_err.c_str(); }

int32_t ldpc_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int usable = 0;
    for (int d = 0; d < cnt; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
    return usable;
}

// The check-degree bucket of the IRR instantiation (bp_team_kernels.hpp) an irregular graph is decoded with: 8 or 16 --
// the 32-wide straight-line code on generic pointers does not fit the registers.  A graph with a FEW wider checks (the
// tail of a random construction; one wide check of a test) still takes the 16-wide instantiation: those checks stay in
// the slot (the packing never puts them in LDS); up to 32 edges they are updated in two halves (check_update_halves:
// the first half's rows are read twice), beyond that on the O(deg^2) path of every kernel (check_update_any: deg^2 / 2
// divisions instead of 2 deg).  Admitted while the checks of 17 ... 32 edges hold at most an eighth of the edges and the
// O(deg^2) ones cost at most 1 / 32 more divisions than the graph has anyway (2 nnz); otherwise 0: every row in the
// slot, the plain 32-wide team kernel.
static int team_irr_dc_bucket(const std::vector<int> &row_ptr, int s, int64_t nnz)
{
    int max_deg = 0;
    int64_t halves = 0;   // edges of the checks of 17 ... 32 edges
    int64_t wide = 0;     // sum of deg^2 over the checks beyond 32 edges
    for (int i = 0; i < s; ++i) {
        const int deg = row_ptr[(size_t)i + 1] - row_ptr[(size_t)i];
        max_deg = std::max(max_deg, deg);
        if (deg > 32) wide += (int64_t)deg * deg;
        else if (deg > 16) halves += deg;
    }
    if (max_deg <= 8) return 8;
    if (max_deg <= 16) return 16;
    return (halves * 8 <= nnz && wide * 8 <= nnz) ? 16 : 0;      // (deg^2 / 2 against 2 nnz / 32)
}

ldpc_status ldpc_bp_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr,
                           const int64_t *rowval, double per, int64_t max_iters,
                           const ldpc_bp_options *options, ldpc_bp_decoder **out)
{
    if (!out) return fail(LDPC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (s < 0 || n < 0 || nnz < 0) return fail(LDPC_ERR_INVALID_ARGUMENT, "negative dimension");
    if (!colptr || (nnz > 0 && !rowval)) return fail(LDPC_ERR_INVALID_ARGUMENT, "colptr/rowval is NULL");
    if (max_iters < 0 || max_iters > INT32_MAX) return fail(LDPC_ERR_INVALID_ARGUMENT, "max_iters out of range");
    if (s >= INT32_MAX || n >= INT32_MAX || nnz >= (int64_t)INT32_MAX / 64 * 8)
        return fail(LDPC_ERR_UNSUPPORTED, "graph too large for 32-bit edge indexing");
    if (colptr[0] != 0 || colptr[n] != nnz)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "colptr[0] must be 0 and colptr[n] must equal nnz (zero-based CSC)");
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) return fail(LDPC_ERR_INVALID_ARGUMENT, "colptr is not non-decreasing");
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (rowval[k] < 0 || rowval[k] >= s)
                return fail(LDPC_ERR_INVALID_ARGUMENT, "rowval entry outside [0, s)");
            if (k > colptr[j] && rowval[k] <= rowval[k - 1])
                return fail(LDPC_ERR_INVALID_ARGUMENT,
                            "row indices must be strictly ascending inside each column (SparseMatrixCSC invariant)");
        }
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(LDPC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    int device = options ? options->device : -1;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(LDPC_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LDPC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if (ldpc_detail::device_stalled(device)) return ldpc_detail::stalled_error(device);

    ldpc_bp_decoder *d = new (std::nothrow) ldpc_bp_decoder();
    if (!d) return fail(LDPC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->s = s; d->n = n; d->nnz = nnz; d->max_iters = max_iters; d->per = per;
    d->device = device;
    d->num_cus = prop.multiProcessorCount;

    // sparse(H') (belief_propagation.jl:64): CSR of H, bits ascending inside each check,
    // plus for every CSC edge its position in that check-major order.
    std::vector<int> row_ptr((size_t)s + 1, 0), edge_bit((size_t)std::max<int64_t>(nnz, 1)),
        col_ptr((size_t)n + 1), csc2csr((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t k = 0; k < nnz; ++k) row_ptr[(size_t)rowval[k] + 1]++;
    for (int64_t i = 0; i < s; ++i) {
        d->max_cdeg = std::max(d->max_cdeg, row_ptr[(size_t)i + 1]);
        row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
    }
    {
        std::vector<int> fill(row_ptr.begin(), row_ptr.end() - 1);
        for (int64_t j = 0; j < n; ++j) {
            col_ptr[(size_t)j] = (int)colptr[j];
            d->max_bdeg = std::max(d->max_bdeg, (int)(colptr[j + 1] - colptr[j]));
            for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
                int q = fill[(size_t)rowval[k]]++;
                edge_bit[(size_t)q] = (int)j;
                csc2csr[(size_t)k] = q;
            }
        }
        col_ptr[(size_t)n] = (int)nnz;
    }

    ldpc_status st = LDPC_OK;
    auto upload = [&](DevBuf &b, const std::vector<int> &v) -> ldpc_status {
        ldpc_status r = b.ensure(std::max<size_t>(v.size(), 1) * sizeof(int));
        if (r != LDPC_OK) return r;
        if (hipMemcpy(b.p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            return fail(LDPC_ERR_HIP, "hipMemcpy of the Tanner graph failed");
        }
        return LDPC_OK;
    };
    if ((st = upload(d->row_ptr, row_ptr)) != LDPC_OK || (st = upload(d->edge_bit, edge_bit)) != LDPC_OK ||
        (st = upload(d->col_ptr, col_ptr)) != LDPC_OK || (st = upload(d->csc2csr, csc2csr)) != LDPC_OK) {
        delete d;
        return st;
    }

    // regular graphs whose degree pair has a rows-in-LDS instantiation (every check max_cdeg edges, every bit max_bdeg)
    if (nnz > 0 && nnz == (int64_t)d->max_cdeg * s && nnz == (int64_t)d->max_bdeg * n && team_rows_degrees_ok(d->max_cdeg, d->max_bdeg)) {
        d->h_csc2csr = csc2csr;
        d->rows_dc = d->max_cdeg; d->rows_dv = d->max_bdeg;
    } else if (nnz > 0) {   // (the tables are only built once a plan with teams of >= 3 members is in sight: team_plan())
        d->h_row_ptr = row_ptr; d->h_edge_bit = edge_bit; d->h_col_ptr = col_ptr; d->h_irr_c2r = csc2csr;
        d->irr_dcb = team_irr_dc_bucket(row_ptr, (int)s, nnz);
    }
    if (const char *e = exp_env("LDPC_TEAM_ROWS")) d->team_rows_on = std::atoi(e) != 0;

    // geometry: waves per tile (fixed by the caller or chosen per batch) and the workspace budget
    int wpt = options ? options->waves_per_tile : 0;
    if (wpt != 0 && wpt != 4 && wpt != 8 && wpt != 16) {
        delete d;
        return fail(LDPC_ERR_INVALID_ARGUMENT, "waves_per_tile must be 0, 4, 8 or 16");
    }
    d->wpt_fixed = wpt;
    d->resident_fixed = options ? std::max(options->resident_tiles, 0) : 0;
    d->defer_thresh = options ? options->defer_threshold : 0;
    if (d->defer_thresh > 48 || d->defer_thresh < -1) { delete d; return fail(LDPC_ERR_INVALID_ARGUMENT, "defer_threshold must be -1, 0 or 1..48"); }
    for (auto &row : d->blocks_cache)
        for (int &v : row) v = -1;
    d->llr_exact = options && options->llr_exact != 0;
    d->variant = options ? options->kernel_variant : 0;
    if (d->variant < 0 || d->variant > 4) { delete d; return fail(LDPC_ERR_INVALID_ARGUMENT, "kernel_variant must be 0 ... 4"); }
    if (const char *e = exp_env("LDPC_TEAM_INJECT_FAULT")) d->inject_fault = std::max(1, std::atoi(e));   // (tests, experiments build: 1 = fault word at once, 2 = a member misses the roll call)
    if (const char *e = exp_env("LDPC_TEAM_ROLLCALL_US")) d->rollcall_ticks = (unsigned)std::max(1, std::atoi(e)) * 100u;
    if (const char *e = exp_env("LDPC_TEAM_CACHE_MIB")) d->team_cache = (size_t)std::max(0, std::atoi(e)) << 20;
    if (const char *e = exp_env("LDPC_TEAM_CACHE_KIB")) d->team_cache = (size_t)std::max(0, std::atoi(e)) << 10;   // (tests: persistent teams on small graphs)
    if (const char *e = exp_env("LDPC_TEAM_DYNAMIC")) d->team_dynamic = std::atoi(e) != 0;
    if (const char *e = exp_env("LDPC_TEAM_XCDS")) d->team_xcds = std::max(1, std::min(8, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_PAIRS")) d->team_pairs = std::atoi(e) & 3;   // (bit 1: four bits at once in the variable sweep)
    if (const char *e = exp_env("LDPC_TEAM_AHEAD")) { d->team_ahead = std::max(0, std::min(65, std::atoi(e))); d->team_ahead_set = true; }
    if (const char *e = exp_env("LDPC_TEAM_AHEAD_FROM")) d->team_ahead_from = std::max(1, std::atoi(e));
    if (const char *e = exp_env("LDPC_TEAM_LLR_RAW")) { const int v = std::atoi(e); if (v == 0 || v == 4 || v == 5 || v == 6) d->team_llr_raw = v; }   // (6: a timing probe, LLRs undefined)
    if (d->llr_exact && d->team_llr_raw == 4) d->team_llr_raw = 5;   // (exact LLRs need all of T)
    if (const char *e = exp_env("LDPC_TEAM_LLR_FOOTPRINT")) d->team_llr_footprint = std::atoi(e) != 0;
    if (const char *e = exp_env("LDPC_TEAM_REGS")) d->team_regs = std::max(0, std::min(kTeamRegRows, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_FLIP")) d->team_flip = std::atoi(e) & 3;
    if (const char *e = exp_env("LDPC_TEAM_PRE")) { d->team_pre = std::max(0, std::min(8, std::atoi(e))); d->team_pre_set = true; }
    if (const char *e = exp_env("LDPC_TEAM_STATIC")) d->team_static_quarters = std::max(0, std::min(4, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_PER_CU")) d->team_per_cu_want = std::max(1, std::min(3, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_MIN_ROWS")) { d->team_min_rows = std::max<int64_t>(1, std::atoll(e)); d->team_min_rows_set = true; }
    d->team_no_margin = exp_env("LDPC_TEAM_NO_MARGIN") != nullptr;          // (experiments: fill the CUs exactly)
    d->team_always_release = exp_env("LDPC_TEAM_ALWAYS_RELEASE") != nullptr;
    d->team_coop = exp_env("LDPC_TEAM_COOP_LAUNCH") != nullptr;
    if (const char *e = exp_env("LDPC_TEAM_MAX")) { d->team_max = std::max(1, std::min(kTeamMaxMembers, std::atoi(e))); d->team_max_set = true; }
    {
        void *fp = nullptr, *fd = nullptr;
        if (hipHostMalloc(&fp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(&fd, fp, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (fp) (void)hipHostFree(fp);
            delete d;
            return fail(LDPC_ERR_OUT_OF_MEMORY, "hipHostMalloc of the fault word failed");
        }
        std::memset(fp, 0, 64);
        d->team_fault = (unsigned int *)fp;
        d->team_fault_dev = (unsigned int *)fd;
    }
    d->lds_logS[0] = lds_logS(s, n, nnz, false);
    d->lds_logS[1] = lds_logS(s, n, nnz, true);
    if (const char *e = exp_env("LDPC_LDS_LOGS")) {   // tuning experiments only
        const int l = std::atoi(e);
        for (int q = 0; q < 2; ++q)
            if (d->lds_logS[q] >= 0 && l >= 0 && l <= 6 &&
                lds_bytes_needed((int)s, (int)n, (int)nnz, 1 << l, q == 1) + 256 <= (size_t)160 * 1024)
                d->lds_logS[q] = l;
    }
    if (d->variant == 2 && (d->lds_logS[0] < 0 || d->lds_logS[1] < 0)) {
        delete d;
        return fail(LDPC_ERR_UNSUPPORTED, "kernel_variant 2 (LDS-resident) requested but the edge messages do not fit the LDS");
    }
    d->node_ok = node_lds_bytes((int)s, (int)n) + 1024 <= (size_t)160 * 1024;
    d->node_msg_lds = d->node_ok && node_lds_bytes((int)s, (int)n) + (size_t)nnz * sizeof(double) + 1024 <= (size_t)160 * 1024;
    if (const char *e = exp_env("LDPC_NODE_MSG_LDS")) d->node_msg_lds = d->node_msg_lds && std::atoi(e) != 0;
    if (d->node_ok && !d->node_msg_lds && !(exp_env("LDPC_NODE_HYBRID") && std::atoi(exp_env("LDPC_NODE_HYBRID")) == 0)) {
        // hybrid: as many leading checks as fit keep their messages in LDS
        size_t room = (size_t)160 * 1024 - 1024 - node_lds_bytes((int)s, (int)n);
        const char *room_env = exp_env("LDPC_NODE_LDS_ROOM");   // tests: bytes of LDS the messages may take
        if (room_env) room = std::min(room, (size_t)std::atoll(room_env));
        int64_t k = 0;
        while (k < s && (size_t)row_ptr[(size_t)k + 1] * sizeof(double) <= room) ++k;
        if (k > 0 && (room_env || (size_t)row_ptr[(size_t)k] * sizeof(double) >= (size_t)16 * 1024)) {
            d->node_split_check = (int)k;
            d->node_split_edge = row_ptr[(size_t)k];
        }
    }
    if (d->variant == 3 && !d->node_ok) {
        delete d;
        return fail(LDPC_ERR_UNSUPPORTED, "kernel_variant 3 (node-parallel) requested but s + n bytes do not fit the LDS");
    }
    // The tile kernel needs ~one 64-syndrome tile per CU before it beats one workgroup per syndrome
    // (measured crossovers in DESIGN.md); LDPC_NODE_MAX_BATCH overrides for experiments.
    d->node_max_batch = (int64_t)d->num_cus * 8;
    if (const char *e = exp_env("LDPC_NODE_MAX_BATCH")) d->node_max_batch = std::atoll(e);
    if (const char *e = exp_env("LDPC_DEFER_T0")) d->defer_t0 = std::max(1, std::min(48, std::atoi(e)));
    if (const char *e = exp_env("LDPC_DEFER_T1")) d->defer_t1 = std::max(0, std::min(48, std::atoi(e)));
    if (const char *e = exp_env("LDPC_DEFER_CAP_TILES")) d->lvl_cap_force = std::max(0, std::atoi(e));
    d->node_take_max = (int64_t)d->num_cus * 8;
    if (const char *e = exp_env("LDPC_NODE_TAKE_MAX")) d->node_take_max = std::atoll(e);
    // keep the workspace inside a sane share of HBM (slots are nnz*512 B each)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        d->ws_budget = free_b / 2;
    } else {
        (void)hipGetLastError();
        d->ws_budget = (size_t)64 << 30;
    }
    // the message workspace itself is allocated on first use, sized min(resident, tiles in the batch)

    if ((st = d->ctrl.ensure(64 * ldpc_bp_decoder::kRing)) != LDPC_OK) { delete d; return st; }
    if (hipMemset(d->ctrl.p, 0, 64 * ldpc_bp_decoder::kRing) != hipSuccess) { (void)hipGetLastError(); delete d; return fail(LDPC_ERR_HIP, "hipMemset failed"); }
    for (bool &c : d->ctrl_clean) c = true;
    if ((st = d->done_ctr.ensure(64)) != LDPC_OK) { delete d; return st; }
    if (hipMemset(d->done_ctr.p, 0, 64) != hipSuccess) { (void)hipGetLastError(); delete d; return fail(LDPC_ERR_HIP, "hipMemset failed"); }
    for (auto &slot : d->ev)
        for (hipEvent_t &e : slot) {
            if (hipEventCreate(&e) != hipSuccess) {
                (void)hipGetLastError();
                delete d;
                return fail(LDPC_ERR_HIP, "hipEventCreate failed");
            }
        }
    *out = d;
    return LDPC_OK;
}

ldpc_status ldpc_bp_destroy(ldpc_bp_decoder *dec)
{
    if (!dec) return LDPC_OK;
    DeviceGuard guard;
    (void)hipSetDevice(dec->device);
    const ldpc_status st = ldpc_detail::wait_device(dec->device, "ldpc_bp_destroy (device synchronise)");
    dec->device_idle = st == LDPC_OK;
    delete dec;          // (a stalled device: the handle goes, what the device may still use is leaked)
    return st;
}

ldpc_status ldpc_bp_get_info(const ldpc_bp_decoder *d, ldpc_bp_info *info)
{
    if (!d || !info) return fail(LDPC_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(info, 0, sizeof *info);
    info->s = d->s; info->n = d->n; info->nnz = d->nnz; info->max_iters = d->max_iters; info->per = d->per;
    info->max_check_degree = d->max_cdeg; info->max_bit_degree = d->max_bdeg;
    info->device = d->device; info->tile_syndromes = kTile; info->waves_per_tile = d->last_threads / 64;
    info->resident_tiles = d->last_grid;
    const DevBuf *all[] = {&d->row_ptr, &d->edge_bit, &d->col_ptr, &d->csc2csr, &d->msg, &d->ctrl, &d->synmask,
                           &d->nevermask, &d->errmask, &d->finmask, &d->llr_t, &d->st_all, &d->node_msg, &d->pipe_dev[0], &d->pipe_dev[1], &d->pipe_dev[2],
                           &d->lvl_state[0], &d->lvl_state[1], &d->lvl_list[0], &d->lvl_list[1], &d->lvl_it[0], &d->lvl_it[1],
                           &d->lvl_syn[0], &d->lvl_syn[1], &d->lvl_never[0], &d->lvl_never[1], &d->lvl_err[0], &d->lvl_err[1], &d->lvl_fin[0], &d->lvl_fin[1],
                           &d->lvl_llr[0], &d->lvl_llr[1], &d->team_ws, &d->team_ws_lvl[0], &d->team_ws_lvl[1],
                           &d->rows_ctab, &d->rows_vtab, &d->rows_lds_edge, &d->rows_reg_edge, &d->rows_posmap,
                           &d->irr_ctab, &d->irr_ptab, &d->irr_ploc, &d->irr_lds_edge, &d->irr_posmap};
    for (const DevBuf *b : all) info->workspace_bytes += (int64_t)b->cap;
    info->last_kernel = d->last_kernel;
    info->last_team_size = d->last_team;
    info->last_lds_rows = d->last_lds_rows;
    info->last_rows_on_chip = d->last_rows_on_chip;
    return LDPC_OK;
}

}  // extern "C"

// Launch of a team grid (bp_team_kernels.hpp): every member of a team must be resident at once, so
//   * the host sizes the grid from the instantiation's occupancy (team_geometry(): at most that many
//     workgroups per CU, with a margin where the registers admit more than one), and
//   * no two team grids of this process run at the same time on one device: a team launch first makes its
//     stream wait for the previous team grid's completion event and then records its own -- two decoders on
//     two streams cannot leave each other's teams half resident.
// That is what hipLaunchCooperativeKernel promises too, and the first version used it; but a process that has
// made ONE cooperative launch crashes inside exit() under rocprofv3 (SIGSEGV in the runtime's exit handler
// after the tool has finalised: profiles/README.md, round 2), and the launch costs 15-19 us more.  A plain
// launch of the same grid has the same residency (MI355X_MICROARCH.md, "Residency and cooperative launch").
// LDPC_TEAM_COOP_LAUNCH=1 brings the cooperative launch back (experiments).
// (Two builds of this library in one process -- the product and the experiments build, which the Python host loads
// side by side -- would each keep an event table of their own and could put two team grids on one device together;
// the host lets the second build ADOPT the first one's table: ldpc_debug_adopt_process_state().)
namespace {
struct ProcessState {
    std::mutex team_mu;
    hipEvent_t team_ev[64] = {};   // per device: completion of the most recent team grid (process lifetime)
};
ProcessState g_own_state;
ProcessState *g_ps = &g_own_state;
}  // namespace
extern "C" void *ldpc_debug_process_state(void) { return g_ps; }
extern "C" ldpc_status ldpc_debug_adopt_process_state(void *state)
{
    if (!state) return fail(LDPC_ERR_INVALID_ARGUMENT, "state is NULL");
    g_ps = (ProcessState *)state;   // (same sources, same layout: both builds come out of one make)
    return LDPC_OK;
}

// Dynamic LDS a team workgroup asks for although it uses none (LDPC_TEAM_LDS_KIB): a way to bound how many
// members the dispatcher can put on one CU (81+ KiB: one, 54+ KiB: two), so that a grid of exactly that many
// members per CU lands evenly.
static size_t team_lds_bytes()
{
    static const size_t v = [] { const char *e = exp_env("LDPC_TEAM_LDS_KIB"); return e ? (size_t)std::max(0, std::min(160, std::atoi(e))) * 1024 : (size_t)0; }();
    return v;
}

static hipError_t launch_team_grid(ldpc_bp_decoder *d, team_kernel_t tk, int grid, void **args, hipStream_t stream, size_t lds)
{
    if (d->team_coop) return hipLaunchCooperativeKernel((const void *)tk, dim3((unsigned)grid), dim3(LDPC_TEAM_THREADS), args, (unsigned)lds, stream);
    std::lock_guard<std::mutex> lk(g_ps->team_mu);
    hipEvent_t *gev = (d->device >= 0 && d->device < 64) ? &g_ps->team_ev[d->device] : nullptr;
    if (gev) {
        if (!*gev) {
            if (hipEventCreateWithFlags(gev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); *gev = nullptr; }
        } else {
            const hipError_t w = hipStreamWaitEvent(stream, *gev, 0);
            if (w != hipSuccess) return w;
        }
    }
    hipError_t e = hipLaunchKernel((const void *)tk, dim3((unsigned)grid), dim3(LDPC_TEAM_THREADS), args, lds, stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess && gev && *gev) e = hipEventRecord(*gev, stream);
    return e;
}

// Latency mode (see ldpc_bp_decoder::lat_pin): the word the last workgroup raises, and its value.
struct LatencyCtl {
    unsigned int *flag;
    unsigned int ticket;
};

// Which kernel a batch goes to (shared by the device entry and the host entry's latency path).
static bool takes_lds_kernel(const ldpc_bp_decoder *d, bool want_llr)
{
    return d->variant != 1 && d->variant != 3 && d->variant != 4 && d->lds_logS[want_llr ? 1 : 0] >= 0;
}
constexpr int64_t kTeamMinRowsOne = 1100;   // message rows per sweep a member of the ONLY team of an XCD must have (team_geometry())

// Workgroups per tile the team kernel (bp_team_kernels.hpp) would use for this batch; 1 = it does not apply.
// Teams are formed inside one XCD (1/8 of the CUs), two workgroups per CU so that every member is
// resident with room to spare (the register budget admits three), and a member should have >= 2048
// message rows per sweep to pay for the three team barriers of an iteration.
// (per_xcd: team workgroups one XCD hosts; gcap: members per tile at most; false = no teams for this decoder)
static bool team_geometry(ldpc_bp_decoder *d, bool want_llr, int *per_xcd, int *gcap, int *gcap_one = nullptr)
{
    if (!(d->variant == 0 || d->variant == 4) || d->team_max < 2 || d->wpt_fixed || d->resident_fixed || d->nnz <= 0) return false;
    const int per_cu_want = d->team_per_cu_want;
    // what this instantiation's registers admit (the wide-degree buckets: one 8-wave workgroup per CU), less one
    // as the margin where that leaves at least one
    int occ = 0;
    if (d->prepare_kernel((const void *)pick_team_kernel(d->max_cdeg, d->max_bdeg, want_llr), LDPC_TEAM_THREADS, team_lds_bytes(), &occ) != LDPC_OK) return false;
    const bool no_margin = d->team_no_margin;
    const int per_cu = no_margin ? std::min(per_cu_want, occ) : std::min(per_cu_want, occ >= 2 ? occ - (occ > per_cu_want ? 0 : 1) : occ);
    if (per_cu < 1) return false;
    const int64_t min_rows = d->team_min_rows;
    *per_xcd = per_cu * (d->num_cus / 8);
    *gcap = (int)std::min<int64_t>(d->team_max, std::max<int64_t>(1, d->nnz / min_rows));   // (LDPC_TEAM_MIN_ROWS=1: tests put teams on tiny graphs)
    // ONE persistent team per XCD may have more members than that -- ALL the CUs of its XCD, or it stays as it is --
    // down to 1100 rows a member (round 3, 16,384 syndromes x 50 iterations: (3,6) n = 16380 with 32 members of 1535
    // rows 132 ms, with 23 of 2136 rows 156 ms; (4,8) n = 12288 / 11264 / 10240 / 9216 with 32 members 139 / 126 / 117 /
    // 105 ms against 155 / 149 / 152 / 110 ms under the 2048-row rule; (3,6) n = 12288 100 against 112 ms; but n = 8192,
    // 1024 rows a member: 96 ms against 90 ms for sixteen teams of 16 -- profiles/r03_minrows_ab.txt, r03_minrows2.txt)
    const bool min_rows_set = d->team_min_rows_set;
    if (gcap_one) {
        const int64_t full = std::min<int64_t>(d->team_max, d->num_cus / 8);
        *gcap_one = (!min_rows_set && d->nnz / kTeamMinRowsOne >= full) ? (int)std::max<int64_t>(*gcap, full) : *gcap;
    }
    return *per_xcd >= 1;
}

// LDS rows a member of a persistent team can hold: 312 x 512 B = 156 KiB of the 160 KiB (the kernel's own few words beside)
constexpr int kTeamRowsMax = 312;

static bool team_rows_possible(const ldpc_bp_decoder *d) { return d->team_rows_on && !d->h_csc2csr.empty() && !d->wpt_fixed; }
static bool team_irr_possible(const ldpc_bp_decoder *d) { return d->team_rows_on && !d->h_row_ptr.empty() && !d->wpt_fixed && d->irr_dcb > 0; }

// What the team plan depends on (pure data: ldpc_debug_team_plan() plans without a device for a CPU test)
struct TeamPlanIn {
    int64_t nnz = 0, max_iters = 0;
    size_t cache = 0;          // budget of message slots in flight (ldpc_bp_decoder::team_cache)
    int xcds_forced = 0;       // LDPC_TEAM_XCDS
    bool team_max_set = false; // LDPC_TEAM_MAX given: no teams over all XCDs for <= 3 tiles
    bool rows_possible = false;
    int rows_dv = 4;           // bit degree of the regular graph (one edge per bit is a candidate for a row in LDS)
    int reg_rows = 0;          // rows a member's waves keep in registers on top of the LDS (W x regs per wave)
    int num_cus = 256;
    int per_xcd = 0, gcap = 0; // team_geometry(): team workgroups one XCD hosts, members per team at most
    int gcap_one = 0;          // ... members at most of a persistent team that has an XCD to itself (>= gcap)
    // <= 3 tiles, members over all XCDs: members at most, message rows a member at least.  (Round 3, one tile of the C3 code,
    // a single decode! / 50 iterations: 64 members 0.294 / 3.44 ms, 96: 0.246 / 2.96, 128: 0.221 / 2.62, 192: 0.230 / 2.70,
    // 256: 0.244 / 3.00; n = 32768, 50 iterations: 64: 5.64, 128: 3.83, 192: 3.57, 256: 3.72 ms -- profiles/r03_scatter_tune.txt)
    int scatter_max = 192, scatter_rows = 512;
    // ... and up to how many tiles a batch is dealt that way (LDPC_TEAM_SCATTER_TILES).  (50 iterations, C3 code / n = 32768,
    // all-XCD teams against one team per XCD: 1 tile 2.4 / 3.4 against 5.1 / 10.2 ms, 2 tiles 3.3 / 5.2, 3 tiles 4.3 / 7.1,
    // 4 tiles 5.09 / 10.7 against 5.00 / 10.2, 5 tiles 6.1 / 13.9 against 5.1 / 12.1 -- profiles/r03_scatter_tiles.txt)
    int scatter_tiles = 3;
    // bytes a team keeps rewriting besides its message slot: with LLRs wanted the posterior odds of every bit of the tile in
    // hand, n x 512 B per iteration (they live in the cache with the slot, and count against the same budget)
    size_t extra = 0;
    // an irregular graph whose tables (team_irr_tables()) were built for teams of irr_G members keep irr_on_chip of the
    // tile's rows in LDS: counted off the slots of such teams
    bool irr_possible = false;
    int irr_G = 0, irr_on_chip = 0;
    // WIDE teams (round 4): T < 8 persistent teams of (all workgroups) / T members each, dealt over ALL XCDs, rows on
    // chip -- for graphs of which eight slots do not fit the Infinity Cache but a few do (n = 65536: 128 MiB a slot, 96 MiB
    // with a quarter of the rows on chip: two; n = 32768: four).  Every barrier then writes the XCDs' L2s back (the
    // members share no L2), and still -- 16,384 syndromes x 50 iterations, profiles/r04_wide_teams.txt -- n = 65536: 910 ms
    // against the tile kernel's 1238 and eight one-XCD teams' 1228 (T = 1: 1112, T = 4: 1133: four slots are 384 MiB);
    // n = 32768: T = 4 442 ms against eight one-XCD teams' 496 (T = 2: 533).  Without rows on chip: no (n = 65536, T = 2:
    // 1300 ms).  0 = automatic (team_wide_auto()), -1 = never, T > 0 = forced (LDPC_TEAM_WIDE, experiments).
    int wide = 0;
};

// What a member is expected to keep on chip (a bit dealt to a member that owns one of its dv checks: 1 / dv of the edges
// are candidates; the LDS holds kTeamRowsMax of them, the waves' registers reg_rows more).
static int team_rows_expected(const TeamPlanIn &in, int G)
{
    if (!in.rows_possible && in.irr_possible) return G == in.irr_G ? in.irr_on_chip / std::max(G, 1) : 0;
    return in.rows_possible ? (int)std::min<int64_t>(kTeamRowsMax + in.reg_rows, in.nnz / std::max(in.rows_dv, 1) / std::max(G, 1)) : 0;
}

// The tables of TeamRows for teams of G members (kept until another G is asked for), for a regular graph whose checks
// have dc edges each and whose bits dv.  Checks are dealt as the kernel deals them -- chunk c of 2 checks to member
// c % G -- and so are the POSITIONS of the bit order, in chunks of 4; the bits are put into positions by the graph: a
// bit goes to the member, among the owners of its dv checks, that has most room left (any member once those are
// full).  Every edge whose check and bit then share the owner is a candidate for a row ON CHIP:
//   * in the REGISTERS of one wave (TeamRegPlan; regs_per_wave > 0).  Inside a member the first `static_c` check
//     chunks and the first `static_v` position chunks of every sweep belong to its waves by right (chunk l of the
//     member's share to wave l % W; the rest is dealt from a counter as the waves finish).  A bit whose owning check
//     sits in a static chunk of wave w is put into a static position of that same wave, as long as the wave has
//     register rows and static positions left: that edge is then read and written by ONE wave in both sweeps of every
//     iteration and lives in its registers, numbered per wave in check order;
//   * else in the member's LDS: up to kTeamRowsMax per member, numbered in check order.
// Tables: ctab [s][4] = per check {mask of its edges in LDS, LDS row of the first of them, mask of its edges in
// registers, register row of the first of them} (a check's rows of either kind follow each other); vtab [n][vt]
// (vt = team_vtab_words(dv)) = per position the CSR rows of the bit's dv edges, where each lives (>= 0: that LDS row,
// -1: the slot, <= -2: register row -2 - x of the wave), the bit (| 1 << 31 when one of its edges is not in the
// slot), padding; lds_edge [G][R], reg_edge [G][W][regs_per_wave] = the CSR rows held (-1: none), for the write-back
// before a hand-off.
// (pure host code: ldpc_debug_team_rows() hands the tables to a CPU test)
struct TeamRegPlan {
    int regs_per_wave = 0;        // 0 = no rows in registers
    int static_c = 0, static_v = 0;   // chunks of a member's share of the check / variable sweep that belong to waves by right (multiples of W)
    int W = LDPC_TEAM_THREADS / 64;
    bool concentrate = true;       // bits go to the owner of their first check where there is room (team_rows_tables())
    bool whole_checks = false;     // ... and only checks with ALL their rows on chip keep them there
    bool strays_last = true;       // bits with a later edge on chip take a member's last positions (team_rows_tables(); LDPC_TEAM_STRAYS_LAST=0: by number)
};
struct TeamRowTables {
    int R = 0;                    // LDS rows per member (the largest count; kTeamRowsMax at most)
    int vt = 0;                   // words per position record of vtab
    size_t in_lds = 0, in_regs = 0;   // edges with a row in LDS / in registers
    std::vector<int> vtab, ctab, lds_edge, reg_edge;
};
// the static part of a member's share under a plan: `frac_num / 4` of the smallest member's chunks, whole rounds of W
static TeamRegPlan team_reg_plan(int n, int s, int G, int regs_per_wave, int quarters, int dv, int dc)
{
    TeamRegPlan rp;
    // Rows of checks that are only PARTLY on chip (a member's room beyond the whole checks of its block: 2-3 % of the
    // rows) put those checks and their bits on the general per-edge updates.  Round 4, 65,536 syndromes x 50 iterations,
    // alternating on one box, with them / whole checks only: bit degree 3 -- (3,6) n = 16380 505.2 / 491.2 ms, (3,9)
    // 503.0 / 498.5 ms: whole checks only; bit degree 4 and 5 -- C3 703.6 / 706.5 ms, (4,10) 354.5 / 358.2 ms, (5,10)
    // 488.9 / 493.4 ms: with them.  (profiles/r04_tform_ab.txt)
    rp.whole_checks = dv == 3;
    // ... and where they stay, their bits take a member's last positions (team_rows_tables()): C3 full-50 702.5 -> 700.9 ms, per
    // 0.02 49.1 -> 48.7 ms, with LLRs 739.8 -> 735.2 ms; check degree 10 lost by it ((4,10) 353.2 -> 354.5 ms, (5,10) 488.5 ->
    // 491.4 ms) and keeps them dealt by number.
    rp.strays_last = dc <= 9;
    const int W = rp.W;
    const int nch_c = (s + kTeamCheckChunk - 1) / kTeamCheckChunk, nch_v = (n + 3) / 4;
    const int min_c = nch_c / G, min_v = nch_v / G;           // the smallest member's share
    rp.static_c = std::max(W, min_c * quarters / 4 / W * W);
    rp.static_v = std::max(W, min_v * quarters / 4 / W * W);
    if (min_c < 2 * W || min_v < 2 * W || quarters <= 0) { rp.static_c = W; rp.static_v = W; }   // (round 2's dealing: a wave's first chunk is its by right)
    else rp.regs_per_wave = regs_per_wave;
    return rp;
}
static TeamRowTables team_rows_tables(int n, int s, int nnz, int dc, int dv, const std::vector<int> &c2r, int G, const TeamRegPlan &rp)
{
    TeamRowTables out;
    const int vt = team_vtab_words(dv), W = rp.W, RC = rp.regs_per_wave;
    auto check_owner = [&](int i) { return (i / kTeamCheckChunk) % G; };
    // the wave that owns check i by right inside its member, or -1 (dealt dynamically)
    auto check_wave = [&](int i) { const int l = (i / kTeamCheckChunk) / G; return (RC > 0 && l < rp.static_c) ? l % W : -1; };
    auto pos_wave = [&](int p) { const int l = (p / 4) / G; return (RC > 0 && l < rp.static_v) ? l % W : -1; };
    std::vector<int> cap((size_t)G, 0), member_of_bit((size_t)n, -1);
    for (int p = 0; p < n; ++p) cap[(size_t)((p / 4) % G)]++;
    std::vector<int> room = cap;
    // A bit goes to the owner of its FIRST check while that member has room, else to the owner of another of its checks
    // (the one with most room).  First check first: in a Gallager code the first block's check i holds bits
    // wr * i ... wr * i + wr - 1, so all of them land with that check's owner and the rows that end up on chip are
    // whole checks' worth -- a quarter of the checks need no memory at all and the others none of the detours of a
    // mixed update (two clean checks of a chunk load their rows together) -- instead of one or two rows in nearly
    // every check (LDPC_TEAM_CONCENTRATE=0, experiments build: most room only, as in round 2).
    for (int j = 0; j < n; ++j) {
        int best = -1;
        for (int k = 0; k < dv; ++k) {
            const int m = check_owner(c2r[(size_t)dv * j + k] / dc);
            if (room[(size_t)m] <= 0) continue;
            if (k == 0 && rp.concentrate) { best = m; break; }
            if (best < 0 || room[(size_t)m] > room[(size_t)best]) best = m;
        }
        if (best >= 0) { member_of_bit[(size_t)j] = best; room[(size_t)best]--; }
    }
    for (int j = 0, m = 0; j < n; ++j) {
        if (member_of_bit[(size_t)j] >= 0) continue;
        while (m < G && room[(size_t)m] == 0) ++m;
        if (m >= G) std::abort();   // (cannot happen: the members' rooms add up to n positions and every bit takes one)
        member_of_bit[(size_t)j] = m; room[(size_t)m]--;
    }
    // positions of every member in ascending order, and which of them belong to a wave by right
    std::vector<std::vector<int>> pos_of((size_t)G), bits_of((size_t)G);
    for (int p = 0; p < n; ++p) pos_of[(size_t)((p / 4) % G)].push_back(p);
    for (int j = 0; j < n; ++j) bits_of[(size_t)member_of_bit[(size_t)j]].push_back(j);
    std::vector<int> bit((size_t)n, -1), reg_of((size_t)nnz, -1);
    std::vector<std::vector<char>> placed_of((size_t)G);        // per member: which of bits_of[m] already have a position (register rows)
    std::vector<std::vector<int>> reg_rows((size_t)G * W);       // per (member, wave): the CSR rows held in registers
    // (whole checks: a wave's registers take a multiple of dc rows, so that no check is split between registers and LDS --
    //  a split check is all on chip and still pays the general update: 30 of 32 rows a wave for dc = 6 and 10)
    const int RCw = rp.concentrate ? RC / dc * dc : RC;
    std::vector<std::vector<int>> reg_bits((size_t)G * W);      // ... and the bits those rows belong to (they get that wave's static positions)
    auto static_positions = [&](int m) {                         // of each wave of member m, ascending
        std::vector<std::vector<int>> spos((size_t)W);
        for (int p : pos_of[(size_t)m]) { const int w = pos_wave(p); if (w >= 0) spos[(size_t)w].push_back(p); }
        return spos;
    };
    for (int m = 0; m < G; ++m) {
        const std::vector<std::vector<int>> spos = static_positions(m);
        std::vector<char> placed(bits_of[(size_t)m].size(), 0);
        if (RC > 0)
            for (size_t b = 0; b < bits_of[(size_t)m].size(); ++b) {
                const int j = bits_of[(size_t)m][b];
                for (int k = 0; k < dv; ++k) {
                    const int q = c2r[(size_t)dv * j + k], i = q / dc;
                    if (check_owner(i) != m) continue;
                    const int w = check_wave(i);
                    if (w < 0 || (int)reg_rows[(size_t)m * W + w].size() >= RCw || reg_bits[(size_t)m * W + w].size() >= spos[(size_t)w].size()) continue;
                    reg_bits[(size_t)m * W + w].push_back(j);
                    reg_rows[(size_t)m * W + w].push_back(q);
                    placed[b] = 1;
                    break;
                }
            }
        placed_of[(size_t)m] = std::move(placed);   // (positions are given out once the LDS rows are known: below)
    }
    // LDS candidates per member (check and bit share the owner, not in registers), in check order; the first kTeamRowsMax of each get rows
    std::vector<char> is_reg((size_t)nnz, 0);
    for (auto &v : reg_rows) for (int q : v) is_reg[(size_t)q] = 1;
    std::vector<std::vector<int>> cand((size_t)G);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < dv; ++k) {
            const int q = c2r[(size_t)dv * j + k];
            if (!is_reg[(size_t)q] && check_owner(q / dc) == member_of_bit[(size_t)j]) cand[(size_t)member_of_bit[(size_t)j]].push_back(q);
        }
    for (auto &v : cand) { std::sort(v.begin(), v.end()); if ((int)v.size() > kTeamRowsMax) v.resize(kTeamRowsMax); }
    if (rp.whole_checks) {
        // only whole checks stay on chip: a check with SOME rows on chip takes the general update (pointers per edge),
        // which costs more than the few rows save -- in a Gallager code these are the stray edges beyond the first block
        std::vector<int> on_chip((size_t)s, 0);
        for (auto &v : reg_rows) for (int q : v) on_chip[(size_t)(q / dc)]++;
        for (auto &v : cand) for (int q : v) on_chip[(size_t)(q / dc)]++;
        auto partial = [&](int q) { return on_chip[(size_t)(q / dc)] < dc; };
        for (auto &v : reg_rows) v.erase(std::remove_if(v.begin(), v.end(), partial), v.end());
        for (auto &v : cand) v.erase(std::remove_if(v.begin(), v.end(), partial), v.end());
    }
    // The other bits fill the positions that are left, in ascending order -- first the bits with at most their FIRST edge on
    // chip, then the "strays" (a later edge in LDS: the room a member has beyond the whole checks of its block).  A position
    // chunk with one stray in it leaves the four-at-once update for the general one, and dealt by number the strays sat
    // in a third of all chunks (30 ... 52 of a member's 128 at the C3 size, and the members with most were the slowest of
    // every variable sweep); together they fill a tenth, at the upper end of the member's positions, which its waves deal
    // among themselves.
    {
        std::vector<char> in_cand((size_t)nnz, 0);
        for (auto &v : cand) for (int q : v) in_cand[(size_t)q] = 1;
        auto is_stray = [&](int j) {
            bool stray = false;
            if (!rp.strays_last) return false;
            for (int k = 1; k < dv; ++k) stray = stray || in_cand[(size_t)c2r[(size_t)dv * j + k]];
            return stray;
        };
        for (int m = 0; m < G; ++m) {
            // the bits with a row in a wave's registers: that wave's static positions, strays last as well
            const std::vector<std::vector<int>> spos = static_positions(m);
            for (int w = 0; w < W; ++w) {
                size_t at = 0;
                for (int pass = 0; pass < 2; ++pass)
                    for (int j : reg_bits[(size_t)m * W + w])
                        if ((int)is_stray(j) == pass) bit[(size_t)spos[(size_t)w][at++]] = j;
            }
            const std::vector<int> &bm = bits_of[(size_t)m];
            const std::vector<char> &placed = placed_of[(size_t)m];
            std::vector<int> order;
            for (int pass = 0; pass < 2; ++pass)
                for (size_t b = 0; b < bm.size(); ++b) {
                    if (placed[b]) continue;
                    if ((int)is_stray(bm[b]) == pass) order.push_back(bm[b]);
                }
            size_t b = 0;
            for (int p : pos_of[(size_t)m]) {
                if (bit[(size_t)p] >= 0) continue;
                if (b >= order.size()) std::abort();   // (cannot happen: a member has as many bits as positions)
                bit[(size_t)p] = order[b++];
            }
        }
    }
    for (int mw = 0; mw < G * W; ++mw) {
        std::sort(reg_rows[(size_t)mw].begin(), reg_rows[(size_t)mw].end());
        for (int x = 0; x < (int)reg_rows[(size_t)mw].size(); ++x) reg_of[(size_t)reg_rows[(size_t)mw][(size_t)x]] = x;
        out.in_regs += reg_rows[(size_t)mw].size();
    }
    int R = 0;
    for (auto &v : cand) R = std::max(R, (int)v.size());
    R = std::max(R, 1);
    std::vector<int> lds_row_of((size_t)nnz, -1);
    std::vector<int> &lds_edge = out.lds_edge, &reg_edge = out.reg_edge;
    lds_edge.assign((size_t)G * R, -1);
    reg_edge.assign((size_t)G * W * std::max(RC, 1), -1);
    std::vector<int> &vtab = out.vtab, &ctab = out.ctab;
    vtab.assign((size_t)n * vt, 0);
    ctab.assign((size_t)s * 4, 0);
    for (int m = 0; m < G; ++m)
        for (int r = 0; r < (int)cand[(size_t)m].size(); ++r) {
            const int q = cand[(size_t)m][(size_t)r], i = q / dc;
            lds_row_of[(size_t)q] = r;
            lds_edge[(size_t)m * R + r] = q;
            if (ctab[(size_t)4 * i] == 0) ctab[(size_t)4 * i + 1] = r;   // (ascending q: the check's LDS edges follow each other)
            ctab[(size_t)4 * i] |= 1 << (q - dc * i);
        }
    for (int mw = 0; mw < G * W; ++mw)
        for (int x = 0; x < (int)reg_rows[(size_t)mw].size(); ++x) {
            const int q = reg_rows[(size_t)mw][(size_t)x], i = q / dc;
            reg_edge[(size_t)mw * RC + x] = q;
            if (ctab[(size_t)4 * i + 2] == 0) ctab[(size_t)4 * i + 3] = x;
            ctab[(size_t)4 * i + 2] |= 1 << (q - dc * i);
        }
    for (int p = 0; p < n; ++p) {
        const int j = bit[(size_t)p];
        bool any = false;
        for (int k = 0; k < dv; ++k) {
            const int q = c2r[(size_t)dv * j + k];
            vtab[(size_t)p * vt + k] = q;
            const int where = reg_of[(size_t)q] >= 0 ? -2 - reg_of[(size_t)q] : lds_row_of[(size_t)q];
            vtab[(size_t)p * vt + dv + k] = where;
            any = any || where != -1;
        }
        vtab[(size_t)p * vt + 2 * dv] = any ? (j | (int)0x80000000u) : j;
    }
    out.R = R;
    out.vt = vt;
    for (auto &v : cand) out.in_lds += v.size();
    return out;
}

// include/ldpc_mi355x_debug.h: the tables above for a CPU test
extern "C" ldpc_status ldpc_debug_team_rows(int64_t s, int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t members,
                                            int32_t regs_per_wave, int32_t static_quarters, int32_t *degrees, int32_t *shape,
                                            int32_t *vtab, int32_t *ctab, int32_t *lds_edge, int32_t *reg_edge)
{
    if (!colptr || !rowval || !degrees || !shape || !vtab || !ctab || !lds_edge || !reg_edge) return fail(LDPC_ERR_INVALID_ARGUMENT, "NULL argument");
    if (s <= 0 || n <= 0 || members < 1 || members > kTeamMaxMembers || regs_per_wave < 0 || regs_per_wave > kTeamRegRows ||
        static_quarters < 0 || static_quarters > 4)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "bad dimension");
    const int64_t nnz = colptr[n];
    if (nnz <= 0 || nnz % s != 0 || nnz % n != 0 || nnz >= INT32_MAX) return fail(LDPC_ERR_UNSUPPORTED, "not a regular graph");
    const int dc = (int)(nnz / s), dv = (int)(nnz / n);
    if (!team_rows_degrees_ok(dc, dv)) return fail(LDPC_ERR_UNSUPPORTED, "no rows-in-LDS instantiation for this degree pair");
    std::vector<int> fill((size_t)s, 0), c2r((size_t)nnz);
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] - colptr[j] != dv) return fail(LDPC_ERR_UNSUPPORTED, "not a regular graph");
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            const int64_t i = rowval[k];
            if (i < 0 || i >= s || fill[(size_t)i] >= dc) return fail(LDPC_ERR_UNSUPPORTED, "not a regular graph");
            c2r[(size_t)k] = (int)(dc * i + fill[(size_t)i]++);   // bits ascending inside a check, as ldpc_bp_create lays the rows out
        }
    }
    const TeamRegPlan rp = team_reg_plan((int)n, (int)s, members, regs_per_wave, static_quarters, dv, dc);
    const TeamRowTables t = team_rows_tables((int)n, (int)s, (int)nnz, dc, dv, c2r, members, rp);
    degrees[0] = dc; degrees[1] = dv;
    shape[0] = t.vt; shape[1] = t.R; shape[2] = rp.static_c; shape[3] = rp.static_v; shape[4] = rp.regs_per_wave;
    std::memcpy(vtab, t.vtab.data(), t.vtab.size() * sizeof(int));
    std::memcpy(ctab, t.ctab.data(), t.ctab.size() * sizeof(int));
    std::memcpy(lds_edge, t.lds_edge.data(), t.lds_edge.size() * sizeof(int));
    std::memcpy(reg_edge, t.reg_edge.data(), t.reg_edge.size() * sizeof(int));
    return LDPC_OK;
}

// IRREGULAR graphs (round 4): whole checks in the LDS of their owners (bp_team_kernels.hpp, IRR).  Checks are dealt as the
// kernel deals them (chunk c of 2 checks to member c % G), positions of the bit order in chunks of 4.  A check can live
// in its owner's LDS when EVERY one of its bits can be given to that member -- then nobody else ever touches its rows
// in either sweep.  That is a set packing over the checks (two checks that share a bit exclude each other); it is
// taken greedily in check order, within each member's capacity (kTeamRowsMax LDS rows, its share of the positions),
// and only over nodes inside the kernel's register buckets (dcb / dvb).  Tables: ctab2 [s + 1][2], ptab [n + 1][2],
// ploc [nnz], lds_edge [G][R] (the CSR rows held, -1 beyond a member's count), posmap [n] (position of every bit).
// (pure host code: ldpc_debug_team_irr() hands the tables to a CPU test)
struct TeamIrrTables {
    int R = 1;
    size_t in_lds = 0;
    std::vector<int> ctab2, ptab, ploc, lds_edge, posmap;
};
static TeamIrrTables team_irr_tables(int n, int s, int nnz, const std::vector<int> &row_ptr, const std::vector<int> &edge_bit,
                                     const std::vector<int> &col_ptr, const std::vector<int> &c2r, int G, int dcb, int dvb)
{
    TeamIrrTables out;
    auto owner = [&](int i) { return (i / kTeamCheckChunk) % G; };
    std::vector<int> room_pos((size_t)G, 0), next_lds((size_t)G, 0), member_of_bit((size_t)n, -1), lds_base((size_t)s, -1);
    for (int p = 0; p < n; ++p) room_pos[(size_t)((p / 4) % G)]++;
    for (int i = 0; i < s; ++i) {
        const int e0 = row_ptr[(size_t)i], deg = row_ptr[(size_t)i + 1] - e0, m = owner(i);
        if (deg <= 0 || deg > dcb || room_pos[(size_t)m] < deg || next_lds[(size_t)m] + deg > kTeamRowsMax) continue;
        bool ok = true;
        for (int k = 0; k < deg && ok; ++k) {
            const int j = edge_bit[(size_t)e0 + k];
            ok = member_of_bit[(size_t)j] < 0 && col_ptr[(size_t)j + 1] - col_ptr[(size_t)j] <= dvb;
        }
        if (!ok) continue;
        for (int k = 0; k < deg; ++k) member_of_bit[(size_t)edge_bit[(size_t)e0 + k]] = m;
        room_pos[(size_t)m] -= deg;
        lds_base[(size_t)i] = next_lds[(size_t)m];
        next_lds[(size_t)m] += deg;
        out.in_lds += (size_t)deg;
    }
    for (int j = 0, m = 0; j < n; ++j) {             // the other bits: wherever there is room
        if (member_of_bit[(size_t)j] >= 0) continue;
        while (m < G && room_pos[(size_t)m] == 0) ++m;
        if (m >= G) std::abort();                    // (cannot happen: the rooms add up to n positions)
        member_of_bit[(size_t)j] = m; room_pos[(size_t)m]--;
    }
    // positions of every member in ascending order take its bits in ascending order
    std::vector<std::vector<int>> bits_of((size_t)G);
    for (int j = 0; j < n; ++j) bits_of[(size_t)member_of_bit[(size_t)j]].push_back(j);
    std::vector<size_t> taken((size_t)G, 0);
    std::vector<int> bit_at((size_t)n, -1);
    out.posmap.assign((size_t)std::max(n, 1), 0);
    for (int p = 0; p < n; ++p) {
        const int m = (p / 4) % G;
        const int j = bits_of[(size_t)m][taken[(size_t)m]++];
        bit_at[(size_t)p] = j;
        out.posmap[(size_t)j] = p;
    }
    for (int m = 0; m < G; ++m) out.R = std::max(out.R, next_lds[(size_t)m]);
    std::vector<int> check_of((size_t)std::max(nnz, 1), 0);
    for (int i = 0; i < s; ++i)
        for (int e = row_ptr[(size_t)i]; e < row_ptr[(size_t)i + 1]; ++e) check_of[(size_t)e] = i;
    out.ctab2.assign(((size_t)s + 1) * 2, -1);
    for (int i = 0; i <= s; ++i) out.ctab2[(size_t)2 * i] = row_ptr[(size_t)i];
    for (int i = 0; i < s; ++i) out.ctab2[(size_t)2 * i + 1] = lds_base[(size_t)i];
    out.lds_edge.assign((size_t)G * out.R, -1);
    for (int i = 0; i < s; ++i)
        if (lds_base[(size_t)i] >= 0)
            for (int e = row_ptr[(size_t)i]; e < row_ptr[(size_t)i + 1]; ++e)
                out.lds_edge[(size_t)owner(i) * out.R + lds_base[(size_t)i] + (e - row_ptr[(size_t)i])] = e;
    out.ptab.assign(((size_t)n + 1) * 2, 0);
    out.ploc.assign((size_t)std::max(nnz, 1), 0);
    int at = 0;
    for (int p = 0; p < n; ++p) {
        const int j = bit_at[(size_t)p];
        out.ptab[(size_t)2 * p] = at;
        bool any = false;
        for (int k = col_ptr[(size_t)j]; k < col_ptr[(size_t)j + 1]; ++k) {
            const int q = c2r[(size_t)k], i = check_of[(size_t)q];
            if (lds_base[(size_t)i] >= 0) { out.ploc[(size_t)at++] = -1 - (lds_base[(size_t)i] + (q - row_ptr[(size_t)i])); any = true; }
            else out.ploc[(size_t)at++] = q;
        }
        out.ptab[(size_t)2 * p + 1] = any ? (j | (int)0x80000000u) : j;
    }
    out.ptab[(size_t)2 * n] = at;
    return out;
}

// include/ldpc_mi355x_debug.h: the tables above for a CPU test
extern "C" ldpc_status ldpc_debug_team_irr(int64_t s, int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t members,
                                           int32_t dc_bucket, int32_t dv_bucket, int32_t *shape, int32_t *ctab2, int32_t *ptab,
                                           int32_t *ploc, int32_t *lds_edge, int32_t *posmap)
{
    if (!colptr || !rowval || !shape || !ctab2 || !ptab || !ploc || !lds_edge || !posmap) return fail(LDPC_ERR_INVALID_ARGUMENT, "NULL argument");
    if (s <= 0 || n <= 0 || members < 1 || members > kTeamMaxMembers || dc_bucket < 1 || dv_bucket < 1) return fail(LDPC_ERR_INVALID_ARGUMENT, "bad dimension");
    const int64_t nnz = colptr[n];
    if (nnz <= 0 || nnz >= INT32_MAX) return fail(LDPC_ERR_INVALID_ARGUMENT, "bad graph");
    std::vector<int> row_ptr((size_t)s + 1, 0), edge_bit((size_t)nnz), col_ptr((size_t)n + 1), c2r((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        if (rowval[k] < 0 || rowval[k] >= s) return fail(LDPC_ERR_INVALID_ARGUMENT, "rowval entry outside [0, s)");
        row_ptr[(size_t)rowval[k] + 1]++;
    }
    for (int64_t i = 0; i < s; ++i) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
    std::vector<int> fill(row_ptr.begin(), row_ptr.end() - 1);
    for (int64_t j = 0; j < n; ++j) {
        col_ptr[(size_t)j] = (int)colptr[j];
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) { const int q = fill[(size_t)rowval[k]]++; edge_bit[(size_t)q] = (int)j; c2r[(size_t)k] = q; }
    }
    col_ptr[(size_t)n] = (int)nnz;
    const TeamIrrTables t = team_irr_tables((int)n, (int)s, (int)nnz, row_ptr, edge_bit, col_ptr, c2r, members, dc_bucket, dv_bucket);
    shape[0] = t.R; shape[1] = (int32_t)t.in_lds;
    std::memcpy(ctab2, t.ctab2.data(), t.ctab2.size() * sizeof(int));
    std::memcpy(ptab, t.ptab.data(), t.ptab.size() * sizeof(int));
    std::memcpy(ploc, t.ploc.data(), (size_t)nnz * sizeof(int));
    std::memcpy(lds_edge, t.lds_edge.data(), t.lds_edge.size() * sizeof(int));
    std::memcpy(posmap, t.posmap.data(), (size_t)n * sizeof(int));
    return LDPC_OK;
}

static ldpc_status team_irr_build(ldpc_bp_decoder *d, int G)
{
    if (d->irr_G == G) return LDPC_OK;
    const TeamIrrTables t = team_irr_tables((int)d->n, (int)d->s, (int)d->nnz, d->h_row_ptr, d->h_edge_bit, d->h_col_ptr, d->h_irr_c2r, G,
                                            d->irr_dcb, team_bucket_dv(d->max_bdeg));
    auto up = [&](DevBuf &b, const std::vector<int> &v) -> ldpc_status {
        ldpc_status r = b.ensure(std::max<size_t>(v.size() * 4, 4));
        if (r != LDPC_OK) return r;
        if (hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return fail(LDPC_ERR_HIP, "hipMemcpy of the team tables failed"); }
        return LDPC_OK;
    };
    ldpc_status st;
    if ((st = ldpc_detail::wait_device(d->device, "team tables of an irregular graph (device synchronise before the tables are replaced)")) != LDPC_OK) return st;
    if ((st = up(d->irr_ctab, t.ctab2)) != LDPC_OK || (st = up(d->irr_ptab, t.ptab)) != LDPC_OK || (st = up(d->irr_ploc, t.ploc)) != LDPC_OK ||
        (st = up(d->irr_lds_edge, t.lds_edge)) != LDPC_OK || (st = up(d->irr_posmap, t.posmap)) != LDPC_OK)
        return st;
    d->irr_G = G; d->irr_R = t.R; d->irr_on_chip = (int)t.in_lds;
    if (exp_env("LDPC_TEAM_DEBUG"))
        std::fprintf(stderr, "[ldpc] team tables (irregular graph): %d members, %d LDS rows each at most, %zu of %d edges in LDS\n", G, t.R, t.in_lds, (int)d->nnz);
    return LDPC_OK;
}

// include/ldpc_mi355x_debug.h: div_core against `/` on the device (a GPU test)
extern "C" ldpc_status ldpc_debug_div_check(int64_t count, const double *num, const double *den, double *out_core, double *out_ieee)
{
    if (count < 0 || (count > 0 && (!num || !den || !out_core || !out_ieee))) return fail(LDPC_ERR_INVALID_ARGUMENT, "bad argument");
    if (count == 0) return LDPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(LDPC_ERR_NO_DEVICE, "no HIP device available"); }
    const size_t bytes = (size_t)count * sizeof(double);
    double *dv = nullptr;
    HIP_TRY(hipMalloc((void **)&dv, 4 * bytes));
    hipError_t e = hipMemcpy(dv, num, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dv + count, den, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(div_check_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, dv, dv + count, dv + 2 * count, dv + 3 * count, (long long)count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_core, dv + 2 * count, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_ieee, dv + 3 * count, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(dv);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(LDPC_ERR_HIP, std::string("ldpc_debug_div_check: ") + hipGetErrorString(e)); }
    return LDPC_OK;
}

extern "C" ldpc_status ldpc_debug_llr_check(int64_t count, const double *odds, double *out_fast, double *out_lib)
{
    if (count < 0 || (count > 0 && (!odds || !out_fast || !out_lib))) return fail(LDPC_ERR_INVALID_ARGUMENT, "bad argument");
    if (count == 0) return LDPC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return fail(LDPC_ERR_NO_DEVICE, "no HIP device available"); }
    const size_t bytes = (size_t)count * sizeof(double);
    double *dv = nullptr;
    HIP_TRY(hipMalloc((void **)&dv, 3 * bytes));
    hipError_t e = hipMemcpy(dv, odds, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(llr_check_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, dv, dv + count, dv + 2 * count, (long long)count);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_fast, dv + count, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_lib, dv + 2 * count, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(dv);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(LDPC_ERR_HIP, std::string("ldpc_debug_llr_check: ") + hipGetErrorString(e)); }
    return LDPC_OK;
}

static ldpc_status team_rows_build(ldpc_bp_decoder *d, int G)
{
    if (d->rows_G == G) return LDPC_OK;
    // (three quarters of a member's share dealt statically measured 1 % faster than dealing all but the first chunk from
    // the counter -- 873 against 883 ms on the full C3 batch --; rows in registers need it)
    TeamRegPlan rp = team_reg_plan((int)d->n, (int)d->s, G, d->team_regs, d->team_static_quarters, d->rows_dv, d->rows_dc);
    if (const char *e = exp_env("LDPC_TEAM_CONCENTRATE")) { rp.concentrate = std::atoi(e) != 0; rp.whole_checks = std::atoi(e) == 2; }
    if (const char *e = exp_env("LDPC_TEAM_STRAYS_LAST")) rp.strays_last = std::atoi(e) != 0;
    const TeamRowTables t = team_rows_tables((int)d->n, (int)d->s, (int)d->nnz, d->rows_dc, d->rows_dv, d->h_csc2csr, G, rp);
    auto up = [&](DevBuf &b, const std::vector<int> &v) -> ldpc_status {
        ldpc_status r = b.ensure(std::max<size_t>(v.size() * 4, 4));
        if (r != LDPC_OK) return r;
        if (hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return fail(LDPC_ERR_HIP, "hipMemcpy of the team row tables failed"); }
        return LDPC_OK;
    };
    ldpc_status st;
    if ((st = ldpc_detail::wait_device(d->device, "team row tables (device synchronise before the tables are replaced)")) != LDPC_OK) return st;   // (a launch that still reads the tables of another G)
    std::vector<int> posmap((size_t)d->n, 0);
    for (int p = 0; p < (int)d->n; ++p) posmap[(size_t)(t.vtab[(size_t)p * t.vt + 2 * d->rows_dv] & 0x7fffffff)] = p;
    if ((st = up(d->rows_posmap, posmap)) != LDPC_OK) return st;
    if ((st = up(d->rows_ctab, t.ctab)) != LDPC_OK || (st = up(d->rows_vtab, t.vtab)) != LDPC_OK ||
        (st = up(d->rows_lds_edge, t.lds_edge)) != LDPC_OK || (st = up(d->rows_reg_edge, t.reg_edge)) != LDPC_OK)
        return st;
    d->rows_G = G; d->rows_R = t.R; d->rows_on_chip = (int)(t.in_lds + t.in_regs);
    d->rows_regs = rp.regs_per_wave; d->rows_static_c = rp.static_c; d->rows_static_v = rp.static_v;
    {   // the leading chunks of EVERY member's share of the check sweep whose two checks have all their rows on chip (TeamRows::first_c)
        const int full = (1 << d->rows_dc) - 1, W = LDPC_TEAM_THREADS / 64;
        int first = rp.static_c;
        for (int m = 0; m < G; ++m) {
            int l = 0;
            for (; l < first; ++l) {
                const int64_t i = (int64_t)kTeamCheckChunk * ((int64_t)l * G + m);
                bool whole = i + kTeamCheckChunk <= d->s;
                for (int q = 0; whole && q < kTeamCheckChunk; ++q)
                    whole = t.ctab[(size_t)4 * (i + q)] == full || t.ctab[(size_t)4 * (i + q) + 2] == full;
                if (!whole) break;
            }
            first = std::min(first, l);
        }
        d->rows_first_c = first / W * W;
    }
    if (exp_env("LDPC_TEAM_DEBUG"))
        std::fprintf(stderr, "[ldpc] team rows: %d members, %d LDS rows each at most, %zu of %d edges in LDS, %zu in registers (%d per wave at most; static chunks %d / %d)\n",
                     G, t.R, t.in_lds, (int)d->nnz, t.in_regs, rp.regs_per_wave, rp.static_c, rp.static_v);
    return LDPC_OK;
}

// Persistent teams whose message slots in flight stay inside the cache budget (LDPC_TEAM_CACHE_MIB, default 240 of
// the Infinity Cache's 256 MiB): how many XCDs host teams (8, 7 or 6), how many teams each, how many members a
// team -- the combination that gives most workgroups a tile.  For the n = 16384 code (32 MiB a slot) with every row in
// the slot that is SEVEN teams of 32: with an eighth the slots fill the cache to the brim and every team is a fifth
// slower (full batch, 50 iterations, round 2: 1111 ms on 8 XCDs, 1011 ms on 7, 1158 ms on 6); with the rows a member
// keeps on chip counted off the slot (`rows`: 312 in LDS + 8 x 32 in registers) it is EIGHT (round 3: 712 ms).
// false: nothing fits, not even the second tier below.
static bool team_fit(const TeamPlanIn &in, int64_t ntiles, bool rows, int *xcds, int *tpx, int *G)
{
    const int per_xcd = in.per_xcd, gcap = in.gcap;
    const size_t state = std::max<size_t>((size_t)in.nnz, 1) * kTile * sizeof(double);
    const size_t cache = in.cache;
    if (!cache) return false;
    // the combination that gives most workgroups a tile -- but among combinations within 15 % of that, the one that
    // uses most XCDs (their L2s and ports; fewer, larger teams per XCD): x runs downwards, so the first one that
    // qualifies wins.  ((3,6) n = 16380: eight teams of 23 on eight XCDs measured 7.1 TB/s, sixteen of 16 6.3)
    int64_t best = 0;
    // (not fewer than seven XCDs: n = 24576, 48 MiB slots -- six cached teams 382 ms for 16,384 syndromes x 50 iterations,
    //  eight partly cached ones 350 ms; n = 20480: seven cached 283 ms, eight 282 ms -- profiles/r03_midsize_plan.txt)
    const int x_hi = in.xcds_forced ? in.xcds_forced : 8, x_lo = in.xcds_forced ? in.xcds_forced : 7;
    for (int pass = 0; pass < 2; ++pass)
    for (int x = x_hi; x >= x_lo; --x)
        for (int t = 1; t <= per_xcd / 3; ++t) {
            if ((size_t)x * (size_t)t * (state + in.extra - (rows ? state / (size_t)std::max(in.rows_dv, 1) : 0)) > cache) break;   // (1 / dv at most can be in LDS)
            if (t > 1 && (int64_t)x * (t - 1) >= ntiles) break;            // no more teams than tiles
            const int g = std::min(t == 1 ? std::max(gcap, in.gcap_one) : gcap, per_xcd / t);
            if (g < 3) break;
            // rows that the members keep in LDS are not in the cache
            if ((size_t)x * (size_t)t * (state + in.extra - (rows ? (size_t)g * team_rows_expected(in, g) * kTile * sizeof(double) : 0)) > cache) continue;
            const int64_t w = std::min<int64_t>((int64_t)x * t, ntiles) * g;   // workgroups with a tile
            if (pass == 0) { if (w > best) best = w; }
            else if (w * 100 >= best * 85) { *xcds = x; *tpx = t; *G = g; return true; }
        }
    if (best > 0) return true;
    // Slots up to 3.3 x the budget: one team per XCD still pays -- the slots are partly cached or not at all, and a team
    // streams what is not as well as three tile-kernel workgroups per CU do, with 8 slots instead of 768 (round 3, 50
    // iterations, teams against the tile kernel: n = 40960, 80 MiB a slot: 16,384 syndromes 693 against 874 ms, 65,536
    // syndromes 2.76 against 2.94 s; n = 49152: 874 against 1071 ms, 3.51 against 3.70 s).  Beyond that (n = 65536,
    // 128 MiB a slot: 4.96 against 4.85 s) the tile kernel stays.  (profiles/r03_midsize_plan.txt)
    // (Graphs WITHOUT rows on chip of this kind -- irregular ones, degree pairs without an instantiation -- stay inside
    //  the budget: 16,384 syndromes x 50 iterations of an irregular n = 20480 graph, 8 x 35 MiB of slots: eight teams
    //  381-392 ms, the tile kernel 319-328 ms, seven teams with whole checks in LDS 319 ms; n = 32768, 8 x 56 MiB: 570 /
    //  626 ms against 518 ms -- profiles/r04_irregular.txt.  Regular graphs with rows on chip get wide teams long before
    //  this: team_wide_auto().)
    if (!in.xcds_forced && in.rows_possible && (size_t)8 * state <= cache / 10 * 33 && std::min(gcap, per_xcd) >= 3) {
        *xcds = 8; *tpx = 1; *G = std::min(gcap, per_xcd);
        return true;
    }
    return false;
}

// How a batch of fresh tiles is dealt to teams (bp_team_kernels.hpp).  G = 1: no teams for it.
//   * up to 3 tiles: one team per tile, its members dealt over ALL XCDs (scatter), 128-192 of them;
//   * as many tiles as fit one round of teams with the slots at most a quarter over the cache budget: one team per
//     tile (8 tiles of the n = 16384 code: 8 teams once rather than 7 teams twice);
//   * otherwise PERSISTENT teams inside the budget (team_fit()): a team takes tile after tile in its own slot;
//   * nothing fits (larger graphs; LDPC_TEAM_CACHE_MIB=0): round 1's rule -- one team per tile, at most one tile
//     per CU -- and batches of more tiles than CUs stay with the tile kernel, which streams from HBM as well as
//     teams would.
struct TeamPlan {
    int G = 1;            // members per team
    int nteams = 0;       // teams (and message slots)
    int grid = 0;         // workgroups to launch
    int xcds = 8, tpx = 0;   // XCDs that host teams, teams per XCD (not in scatter mode)
    bool scatter = false;
    bool wide = false;       // a few persistent teams over all XCDs (TeamPlanIn::wide)
    bool irr = false;        // an irregular graph: members keep whole checks in LDS (team_irr_tables())
    bool rows = false;       // members keep the rows that only they touch in LDS (TeamRows)
};

// How many wide teams (TeamPlanIn::wide) a batch of ntiles gets by itself: as many as keep their slots -- less the rows
// their members keep on chip, plus a tile's LLR rows when wanted -- inside the cache budget, when that is fewer than the
// eight one-XCD teams the plan would otherwise build (eight or more fit: nothing to do, the C3 code) and the graph has a
// rows-on-chip instantiation (without rows on chip wide teams lose).  0 = none.
static int team_wide_auto(const TeamPlanIn &in, int64_t ntiles)
{
    if (!in.cache || in.xcds_forced || in.team_max_set || !in.rows_possible || in.reg_rows <= 0 || in.per_xcd < 8) return 0;
    const size_t state = std::max<size_t>((size_t)in.nnz, 1) * kTile * sizeof(double);
    for (int T = 7; T >= 1; --T) {
        const int G = (int)std::min<int64_t>(kTeamMaxMembers, (int64_t)8 * in.per_xcd / T);
        if (in.nnz / G < in.scatter_rows) continue;                       // (a member keeps >= 512 message rows per sweep)
        const size_t on_chip = (size_t)G * (size_t)team_rows_expected(in, G) * kTile * sizeof(double);
        const size_t slot = state - std::min(on_chip, state) + in.extra;
        if ((size_t)T * slot > in.cache) continue;
        // eight one-XCD teams' slots would fit as well, or nearly (a quarter over the budget): that plan (no write-backs at
        // the barriers) stays
        const int G8 = std::min(std::max(in.gcap, in.gcap_one), in.per_xcd);
        const size_t slot8 = state - std::min((size_t)G8 * (size_t)team_rows_expected(in, G8) * kTile * sizeof(double), state) + in.extra;
        if ((size_t)8 * slot8 <= in.cache + in.cache / 4) return 0;
        return (int)std::min<int64_t>(T, ntiles);
    }
    return 0;
}

static TeamPlan team_plan_pure(const TeamPlanIn &in, int64_t batch)
{
    TeamPlan pl;
    const int per_xcd = in.per_xcd, gcap = in.gcap;
    const int64_t ntiles = (batch + kTile - 1) / kTile;
    if (ntiles < 1 || per_xcd < 1) return pl;
    if ((size_t)ntiles * ((size_t)in.max_iters + 32) * sizeof(u64) > ((size_t)64 << 20)) return pl;   // mismatch words per tile and iteration
    int64_t team = 1, nteams = 0;
    const int wide = in.wide > 0 ? in.wide : in.wide == 0 ? team_wide_auto(in, ntiles) : 0;
    if (wide > 0 && ntiles > in.scatter_tiles) {
        team = std::min<int64_t>(kTeamMaxMembers, (int64_t)8 * per_xcd / wide);
        nteams = std::min<int64_t>(wide, ntiles);
        pl.scatter = true;
        pl.wide = true;
    } else if (ntiles <= in.scatter_tiles && !in.team_max_set) {
        // dealt over all 8 XCDs (scatter mode of the kernel): larger teams pay -- one tile of the C3 code, 50
        // iterations: 32 members 5.4 ms, 48: 4.3 ms, 64: 3.4 ms, 128: 2.6 ms (a member still has >= 512 message rows per sweep)
        const int64_t cap = std::min<int64_t>(in.scatter_max, std::max<int64_t>(gcap, in.nnz / std::max(in.scatter_rows, 1)));
        team = std::min<int64_t>(cap, (int64_t)8 * per_xcd / ntiles);
        nteams = ntiles;
        pl.scatter = true;
    } else {
        const int64_t need = (ntiles + 7) / 8;
        const size_t state = std::max<size_t>((size_t)in.nnz, 1) * kTile * sizeof(double);
        int x = 8, t = (int)need, g = 0;
        const bool one_round = in.cache && !in.xcds_forced && (size_t)8 * (size_t)need * state <= in.cache + in.cache / 4;
        // (With the rows in LDS alone -- 15 % of a tile -- the budget is applied to whole slots: eight teams of the C3 code,
        // 8 x 27 MiB, measured 981 ms for the full batch, seven 957.  With rows in the waves' registers as well a quarter
        // of a tile is on chip and the rows on chip are taken off the slots: 8 x 24 MiB fit, and eight teams measured
        // 833 ms against 877 on seven -- round 3.)
        if (one_round || !team_fit(in, ntiles, (in.rows_possible && in.reg_rows > 0) || (in.irr_possible && in.irr_on_chip > 0), &x, &t, &g)) {
            if (!one_round && ntiles > in.num_cus) return pl;
            x = 8; t = (int)need;
            g = (int)std::min<int64_t>(gcap, (int64_t)per_xcd / t);
        }
        team = g;
        nteams = (int64_t)x * t;
        pl.xcds = x; pl.tpx = t;
    }
    if (team < 3) return pl;   // two workgroups per tile measured no better than the tile kernel's one of 16 waves
    pl.G = (int)std::min<int64_t>(team, kTeamMaxMembers);
    pl.rows = (!pl.scatter || pl.wide) && in.rows_possible && team_rows_expected(in, pl.G) >= 16;
    pl.irr = !pl.scatter && !in.rows_possible && in.irr_possible;
    pl.nteams = (int)nteams;
    pl.grid = pl.scatter ? pl.nteams * pl.G : 8 * pl.G * pl.tpx;
    return pl;
}

static TeamPlanIn team_plan_in(const ldpc_bp_decoder *d, int per_xcd, int gcap, int gcap_one = 0, bool want_llr = false)
{
    TeamPlanIn in;
    in.extra = (want_llr && d->team_llr_footprint) ? (size_t)std::max<int64_t>(d->n, 0) * kTile * (d->team_llr_raw == 5 ? sizeof(double) : sizeof(unsigned int)) : 0;
    in.nnz = d->nnz; in.max_iters = d->max_iters; in.cache = d->team_cache; in.xcds_forced = d->team_xcds;
    in.team_max_set = d->team_max_set; in.rows_possible = team_rows_possible(d); in.rows_dv = std::max(d->rows_dv, 1); in.num_cus = d->num_cus;
    in.reg_rows = in.rows_possible ? d->team_regs * (LDPC_TEAM_THREADS / 64) : 0;
    in.per_xcd = per_xcd; in.gcap = gcap; in.gcap_one = std::max(gcap, gcap_one);
    if (const char *e = exp_env("LDPC_TEAM_SCATTER_MAX")) in.scatter_max = std::max(3, std::min(kTeamMaxMembers, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_SCATTER_ROWS")) in.scatter_rows = std::max(1, std::atoi(e));
    if (const char *e = exp_env("LDPC_TEAM_SCATTER_TILES")) in.scatter_tiles = std::max(0, std::min(16, std::atoi(e)));
    if (const char *e = exp_env("LDPC_TEAM_WIDE")) in.wide = std::max(-1, std::min(8, std::atoi(e)));
    return in;
}

static TeamPlan team_plan(ldpc_bp_decoder *d, int64_t batch, bool want_llr)
{
    int per_xcd = 0, gcap = 0, gcap_one = 0;
    if (!team_geometry(d, want_llr, &per_xcd, &gcap, &gcap_one)) return TeamPlan();
    TeamPlanIn in = team_plan_in(d, per_xcd, gcap, gcap_one, want_llr);
    if (!in.rows_possible && team_irr_possible(d)) {
        // an irregular graph: its tables are built once for the team size the plan is most likely to choose (all the CUs
        // of an XCD, or the most members the 2048-row rule admits), so that the plan can count the rows in LDS off the slots
        const int G0 = std::min(std::max(gcap, gcap_one), per_xcd);
        if (d->irr_G == 0 && G0 >= 3) (void)team_irr_build(d, G0);
        in.irr_possible = true; in.irr_G = d->irr_G; in.irr_on_chip = d->irr_on_chip;
    }
    return team_plan_pure(in, batch);
}

// include/ldpc_mi355x_debug.h: the plan for a CPU test (an MI355X's geometry: 256 CUs, one team workgroup per CU)
extern "C" ldpc_status ldpc_debug_team_plan(int64_t nnz, int64_t max_iters, int64_t batch, int32_t cache_mib, int32_t rows_dv,
                                            int32_t out[6])
{
    if (!out || nnz < 0 || batch < 0 || cache_mib < 0) return fail(LDPC_ERR_INVALID_ARGUMENT, "bad argument");
    TeamPlanIn in;
    in.nnz = nnz; in.max_iters = max_iters; in.cache = (size_t)cache_mib << 20; in.rows_possible = rows_dv > 0; in.rows_dv = std::max(rows_dv, 1);
    in.reg_rows = rows_dv > 0 ? kTeamRegRows * (LDPC_TEAM_THREADS / 64) : 0;
    in.num_cus = 256; in.per_xcd = 32;
    in.gcap = (int)std::min<int64_t>(32, std::max<int64_t>(1, nnz / 2048));
    in.gcap_one = nnz / kTeamMinRowsOne >= 32 ? 32 : in.gcap;
    const TeamPlan pl = team_plan_pure(in, batch);
    out[0] = pl.G; out[1] = pl.nteams; out[2] = pl.grid; out[3] = pl.xcds; out[4] = pl.scatter ? 1 : 0; out[5] = pl.rows ? 1 : 0;
    return LDPC_OK;
}

static int team_size(ldpc_bp_decoder *d, int64_t batch, bool want_llr) { return team_plan(d, batch, want_llr).G; }

// Small batches on graphs beyond the LDS: one workgroup per syndrome (bp_node_kernels.hpp) or teams of
// workgroups per 64-syndrome tile?  Estimated time of one iteration, from the measurements in DESIGN.md:
//   node kernel: 1.9 ns per edge and syndrome on its CU (the CU's address path), one syndrome per CU at a
//                time, three times that once the message slots in flight outgrow the L2s (32 MiB); 1.07 ns
//                for the edges whose messages the hybrid placement keeps in LDS;
//   team kernel: 41 ns per edge for a tile at one CU's pace, divided among the G members, + ~45 us for the
//                three team barriers (~30 us for the small teams-over-all-XCDs geometry of <= 3 tiles).
// (A partial tile costs the team kernel as much as a full one, so below 64 syndromes this is a contest between
// one syndrome per workgroup and up to 64 workgroups on one tile: n = 16384 and larger go to the team.)
// Where the team kernel does not apply, the node kernel keeps the batches up to node_max_batch.
static bool takes_node_kernel(ldpc_bp_decoder *d, int64_t batch, bool want_llr)
{
    if (takes_lds_kernel(d, want_llr) || !d->node_ok) return false;
    if (d->variant == 3) return true;
    if (d->variant != 0) return false;
    if (d->node_msg_lds) return true;   // messages in LDS: not HBM-bound at any batch size (n = 4096)
    const int G = team_size(d, batch, want_llr);
    if (G < 2) return batch < kTile || batch <= d->node_max_batch;
    const double edges = (double)d->nnz;
    const double rounds = (double)((batch + d->num_cus - 1) / d->num_cus);
    // hybrid placement: the share g of the messages that stays in global memory pays the address-path price, the
    // rest the LDS price (1.07 ns per edge: 17.5 us per iteration at nnz = 16384)
    const double g = d->node_split_check > 0 ? 1.0 - (double)d->node_split_edge / std::max(edges, 1.0) : 1.0;
    const double in_flight = (double)std::min<int64_t>(batch, 2 * (int64_t)d->num_cus) * edges * g * 8.0;
    const double est_node = rounds * (edges * (1.0 - g) * 1.07e-3 + edges * g * 1.9e-3 * (in_flight > 32.0 * 1048576.0 ? 3.0 : 1.0));
    const double est_team = edges * 41e-3 / (double)G + (batch <= 3 * kTile ? 30.0 : 45.0);   // (<= 3 tiles: dealt over all XCDs)
    return est_node < est_team;
}

// A team barrier that timed out (bp_team_kernels.hpp) has written the number of its call into the host-mapped
// fault word; every team grid enqueued since has seen the word and left without results.  Wait for everything
// enqueued on the handle (no member may still be polling the word when it is cleared), clear it, keep teams off
// for this decoder, and say which calls were hit.  LDPC_OK when nothing happened.
static ldpc_status report_team_fault(ldpc_bp_decoder *d)
{
    if (!d->team_fault) return LDPC_OK;
    const unsigned ticket = __atomic_load_n(d->team_fault, __ATOMIC_ACQUIRE);
    if (ticket == 0u) return LDPC_OK;
    if (d->last_ev) {
        (void)hipSetDevice(d->device);
        // (no member may still be polling the word when it is cleared: a wait that expires leaves it set)
        const ldpc_status wst = ldpc_detail::wait_event(d->last_ev, d->device, "team fault report (wait for the handle's last call)");
        if (wst != LDPC_OK) return wst;
    }
    __atomic_store_n(d->team_fault, 0u, __ATOMIC_RELEASE);
    d->team_max = 1;
    // the ticket holds the low 31 bits of the call number (counting from 1); bit 31: the roll call failed
    uint64_t first = (d->ncalls & ~(uint64_t)0x7fffffff) | (uint64_t)(ticket & 0x7fffffffu);
    if (first > d->ncalls && first >= ((uint64_t)1 << 31)) first -= (uint64_t)1 << 31;
    const bool rollcall = (ticket & kTeamRollcallFailed) != 0u;
    return fail(LDPC_ERR_HIP, "call #" + std::to_string(first) + " on this decoder " +
                (rollcall ? "found a team of workgroups incomplete at launch (roll call timed out: the GPU is shared with other work)"
                          : "lost a workgroup of a team (team barrier timed out)") +
                ": the results of that call and of every team-kernel call enqueued after it (up to call #" + std::to_string(d->ncalls) +
                ") are invalid; teams are off for this decoder from here on");
}

static ldpc_status decode_device_impl(ldpc_bp_decoder *d, int64_t batch, const uint8_t *d_syn,
                                      uint8_t *d_err, uint8_t *d_conv, double *d_llr, int32_t *d_iters,
                                      void *stream_v, const LatencyCtl *lat)
{
    if (!d) return fail(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return fail(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    hipStream_t stream = (hipStream_t)stream_v;
    if (batch == 0) return LDPC_OK;
    if ((d->s > 0 && !d_syn) || (d->n > 0 && !d_err) || !d_conv)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "syndromes/errors/converged pointer is NULL");
    HIP_TRY(hipSetDevice(d->device));
    if (ldpc_detail::device_stalled(d->device)) return ldpc_detail::stalled_error(d->device);
    const int64_t s = d->s, n = d->n;
    {
        const ldpc_status fst = report_team_fault(d);   // a refused call is not counted and enqueues nothing
        if (fst != LDPC_OK) return fst;
    }
    // From here on the call has a number and a slot of the timing ring.  If a HIP call fails further down the
    // function returns with the launches made so far enqueued (they only touch this call's own buffers and the
    // handle's workspace), the slot consumed and `timed` false: ldpc_bp_call_timing reports zeros for it.
    const int slot = (int)(d->ncalls++ % ldpc_bp_decoder::kRing);
    hipEvent_t *ev = d->ev[slot];
    char *ctrl = (char *)d->ctrl.p + 64 * slot;
    d->timed[slot] = false;
    d->two_events[slot] = false;
    // single-kernel paths: the kernel zeroes the control slot of the NEXT call, so a call that finds its
    // slot clean enqueues no memset, and only two events bracket the one kernel (measured on C2, batch
    // 4096: fill kernel 3.5 us + ~10 us of dependency gap on either side of a 94 us decode kernel)
    const int nslot = (slot + 1) % ldpc_bp_decoder::kRing;
    char *next_ctrl = (char *)d->ctrl.p + 64 * nslot;
    // Calls on one handle execute in call order whatever streams they are given (they share its workspace and
    // its control slots -- a kernel of call N zeroes the slot of call N+1): a call that arrives on another
    // stream than its predecessor waits for that one's last event first.
    if (d->last_ev && d->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, d->last_ev, 0));
    d->last_stream = stream;

    if (d->max_iters == 0) {
        // the loop at belief_propagation.jl:134 never runs: err = 0, log_probabs = 0, converged = false
        if (n > 0) HIP_TRY(hipMemsetAsync(d_err, 0, (size_t)batch * n, stream));
        HIP_TRY(hipMemsetAsync(d_conv, 0, (size_t)batch, stream));
        if (d_llr && n > 0) HIP_TRY(hipMemsetAsync(d_llr, 0, (size_t)batch * n * sizeof(double), stream));
        if (d_iters) HIP_TRY(hipMemsetAsync(d_iters, 0, (size_t)batch * sizeof(int32_t), stream));
        return LDPC_OK;
    }

    const bool want_llr_early = d_llr != nullptr;
    if (takes_lds_kernel(d, want_llr_early)) {
        // ---- on-chip path: messages never leave the LDS (bp_lds_kernels.hpp)
        const int logS = d->lds_logS[want_llr_early ? 1 : 0];
        const int64_t ngroups64 = (batch + (1ll << logS) - 1) >> logS;
        if (ngroups64 > (1ll << 30)) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one call");
        d->last_kernel = 2; d->last_team = 1; d->last_lds_rows = 0; d->last_rows_on_chip = 0;
        LdsParams lp{};
        lp.s = (int)s; lp.n = (int)n; lp.nnz = (int)d->nnz; lp.max_iters = (int)d->max_iters;
        lp.logS = logS; lp.ngroups = (int)ngroups64; lp.batch = batch;
        lp.r = d->per / (1 - d->per);
        lp.syn = d_syn; lp.err = d_err; lp.conv = d_conv; lp.iters = d_iters; lp.llr = d_llr;
        lp.llr_exact = d->llr_exact ? 1 : 0;
        lp.queue = (unsigned int *)ctrl;
        lp.sum_iters = (u64 *)(ctrl + 8);
        lp.phase_ticks = (u64 *)(ctrl + 16);
        lp.done_count = (unsigned int *)d->done_ctr.p; lp.done_flag = nullptr; lp.done_ticket = 0;
        lp.next_ctrl = nullptr;
        const size_t lds = lds_bytes_needed((int)s, (int)n, (int)d->nnz, 1 << logS, want_llr_early);
        // 512 threads when two or more workgroups share a CU; when the LDS footprint admits only
        // one, give that one all 16 waves (measured +21 % on the (9,10)-regular n=1000 code)
        const bool lone = 2 * (lds + 256) > (size_t)160 * 1024;
        const int lthreads = d->wpt_fixed ? ((d->wpt_fixed == 4) ? 256 : (d->wpt_fixed >= 12 ? 1024 : 512))
                                          : (lone ? 1024 : 512);
        lds_kernel_t lk = pick_lds_kernel(d->max_cdeg, d->max_bdeg, want_llr_early, lthreads);
        int per_cu = 0;
        ldpc_status pst = d->prepare_kernel((const void *)lk, lthreads, lds, &per_cu);
        if (pst != LDPC_OK) return pst;
        const int lgrid = (int)std::min<int64_t>(ngroups64, (int64_t)per_cu * d->num_cus);
        // dequeue in chunks: ~16 dequeues per workgroup keep the load balanced and the queue word quiet
        lp.chunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, ngroups64 / ((int64_t)lgrid * 16)));
        if (lat) {   // one launch and nothing else: no queue, no statistics, no events
            lp.queue = nullptr; lp.sum_iters = nullptr; lp.chunk = 1;
            lp.done_flag = lat->flag; lp.done_ticket = lat->ticket;
            hipLaunchKernelGGL(lk, dim3((unsigned)ngroups64), dim3((unsigned)lthreads), lds, stream, lp,
                               (const int *)d->row_ptr.p, (const int *)d->edge_bit.p, (const int *)d->col_ptr.p,
                               (const int *)d->csc2csr.p);
            HIP_TRY(hipGetLastError());
            return LDPC_OK;
        }
        if (!d->ctrl_clean[slot]) HIP_TRY(hipMemsetAsync(ctrl, 0, 64, stream));
        d->ctrl_clean[slot] = false;
        lp.next_ctrl = (u64 *)next_ctrl;
        HIP_TRY(hipEventRecord(ev[1], stream));
        hipLaunchKernelGGL(lk, dim3((unsigned)lgrid), dim3((unsigned)lthreads), lds, stream, lp,
                           (const int *)d->row_ptr.p, (const int *)d->edge_bit.p, (const int *)d->col_ptr.p,
                           (const int *)d->csc2csr.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[2], stream));
        d->ctrl_clean[nslot] = true;
        d->timed[nslot] = false;
        d->timed[slot] = true;
        d->two_events[slot] = true;
        d->last_ev = ev[2];
        return LDPC_OK;
    }

    if (takes_node_kernel(d, batch, want_llr_early)) {
        // ---- small batch on a large graph: one workgroup per syndrome, one thread per node (bp_node_kernels.hpp)
        if (batch > (1ll << 30)) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one call");
        // all 16 waves of a CU on one syndrome while there are CUs to spare, else two 8-wave workgroups per CU
        const bool mlds = d->node_msg_lds;   // messages in LDS: one 16-wave workgroup per CU
        const int msg_mode = mlds ? 1 : (d->node_split_check > 0 ? 2 : 0);
        const int nthreads = msg_mode ? 1024 : (d->wpt_fixed ? (d->wpt_fixed >= 12 ? 1024 : 512) : (batch <= d->num_cus ? 1024 : 512));
        const size_t nlds = node_lds_bytes((int)s, (int)n) +
                            (msg_mode == 1 ? (size_t)d->nnz : msg_mode == 2 ? (size_t)d->node_split_edge : 0) * sizeof(double);
        node_kernel_t nk = pick_node_kernel(d->max_cdeg, d->max_bdeg, want_llr_early, nthreads, msg_mode);
        int per_cu = 0;
        ldpc_status pst = d->prepare_kernel((const void *)nk, nthreads, nlds, &per_cu);
        if (pst != LDPC_OK) return pst;
        per_cu = std::min(per_cu, 2);
        const int ngrid = lat ? (int)batch : (int)std::min<int64_t>(batch, (int64_t)per_cu * d->num_cus);
        const size_t stride = (std::max<size_t>((size_t)d->nnz, 1) + 63) & ~(size_t)63;   // 512-byte aligned slots
        ldpc_status nst = d->node_msg.ensure(mlds ? 64 : (size_t)ngrid * stride * sizeof(double));
        if (nst != LDPC_OK) return nst;
        d->last_kernel = 3; d->last_team = 1; d->last_lds_rows = 0; d->last_rows_on_chip = 0;
        NodeParams np{};
        np.s = (int)s; np.n = (int)n; np.nnz = (int)d->nnz; np.max_iters = (int)d->max_iters;
        np.batch = batch; np.r = d->per / (1 - d->per);
        np.syn = d_syn; np.err = d_err; np.conv = d_conv; np.iters = d_iters; np.llr = d_llr; np.llr_exact = d->llr_exact ? 1 : 0;
        np.msg = (double *)d->node_msg.p; np.slot_stride = (long long)stride;
        np.queue = (unsigned int *)ctrl;
        np.sum_iters = (u64 *)(ctrl + 8);
        np.index = nullptr; np.count_dev = nullptr; np.count_max = 0;
        np.split_check = d->node_split_check; np.split_edge = d->node_split_edge;
        np.done_count = (unsigned int *)d->done_ctr.p; np.done_flag = nullptr; np.done_ticket = 0;
        np.next_ctrl = nullptr;
        if (lat) {
            np.queue = nullptr; np.sum_iters = nullptr;
            np.done_flag = lat->flag; np.done_ticket = lat->ticket;
            hipLaunchKernelGGL(nk, dim3((unsigned)ngrid), dim3((unsigned)nthreads), nlds, stream, np,
                               (const int *)d->row_ptr.p, (const int *)d->edge_bit.p, (const int *)d->col_ptr.p,
                               (const int *)d->csc2csr.p);
            HIP_TRY(hipGetLastError());
            return LDPC_OK;
        }
        if (!d->ctrl_clean[slot]) HIP_TRY(hipMemsetAsync(ctrl, 0, 64, stream));
        d->ctrl_clean[slot] = false;
        np.next_ctrl = (u64 *)next_ctrl;
        HIP_TRY(hipEventRecord(ev[1], stream));
        hipLaunchKernelGGL(nk, dim3((unsigned)ngrid), dim3((unsigned)nthreads), nlds, stream, np,
                           (const int *)d->row_ptr.p, (const int *)d->edge_bit.p, (const int *)d->col_ptr.p,
                           (const int *)d->csc2csr.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[2], stream));
        d->ctrl_clean[nslot] = true;
        d->timed[nslot] = false;
        d->two_events[slot] = true;
        d->timed[slot] = true;
        d->last_ev = ev[2];
        d->last_threads = nthreads; d->last_grid = ngrid;
        return LDPC_OK;
    }

    if (lat) return fail(LDPC_ERR_HIP, "internal: latency mode reached the tile kernel");
    const int64_t ntiles64 = (batch + kTile - 1) / kTile;
    if (ntiles64 > (1 << 30)) return fail(LDPC_ERR_UNSUPPORTED, "batch too large for one call");
    const int ntiles = (int)ntiles64;
    const bool want_llr = d_llr != nullptr;

    ldpc_status st;
    if ((st = d->synmask.ensure(std::max<size_t>((size_t)ntiles * s, 1) * sizeof(u64))) != LDPC_OK) return st;
    if ((st = d->nevermask.ensure((size_t)ntiles * sizeof(u64))) != LDPC_OK) return st;
    // (two sets of decision words: a team that runs ahead -- bp_team_kernels.hpp -- writes those of odd iterations into the second)
    if ((st = d->errmask.ensure(2 * std::max<size_t>((size_t)ntiles * n, 1) * sizeof(u64))) != LDPC_OK) return st;
    if ((st = d->finmask.ensure(std::max<size_t>((size_t)ntiles * n, 1) * sizeof(u64))) != LDPC_OK) return st;
    if (want_llr && (st = d->llr_t.ensure(std::max<size_t>((size_t)ntiles * (((size_t)n + 3) & ~(size_t)3), 1) * kTile * sizeof(double))) != LDPC_OK)
        return st;
    // Geometry of this launch: 8 waves per tile (three workgroups per CU) is the measured best
    // whenever there are more tiles than CUs -- also against geometries that make every tile
    // resident at once (6 waves x 4 per CU: -10 %); with at most one tile per CU, 16 waves per
    // tile put more of the chip to work.  The caller may fix it.
    int threads = 0, grid = 0;
    {
        const size_t slot_bytes = std::max<size_t>((size_t)d->nnz, 1) * kTile * sizeof(double) + slot_pad_bytes();
        const int max_slots = (int)std::max<size_t>(std::min<size_t>(d->ws_budget / slot_bytes, 1u << 30), 1);
        const int wpt = d->wpt_fixed ? d->wpt_fixed : (ntiles <= d->num_cus ? 16 : 8);
        int &bc = d->blocks_cache[want_llr ? 1 : 0][wpt];
        if (bc < 0) {
            bp_kernel_t kq = pick_kernel(d->max_cdeg, d->max_bdeg, want_llr, wpt * 64);
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)kq, wpt * 64, 0) != hipSuccess || nb <= 0) {
                (void)hipGetLastError();
                nb = 1;
            }
            bc = nb;
        }
        int slots = d->resident_fixed ? d->resident_fixed : bc * d->num_cus;
        slots = std::max(1, std::min(slots, max_slots));
        threads = wpt * 64;
        grid = std::min(slots, ntiles);
    }
    // Teams of workgroups on the tiles (bp_team_kernels.hpp, team_plan()): medium batches, where one workgroup per
    // tile leaves CUs idle, and large ones of graphs whose team slots fit the Infinity Cache.
    // (kernel_variant 4 skips the LDS and node kernels; batches the team kernel cannot take -- an empty graph,
    // slots beyond the cache -- go to the tile kernel)
    const TeamPlan plan = team_plan(d, batch, want_llr);
    const int team = plan.G;
    const int tile_grid = grid, tile_threads = threads;
    if (team > 1) { threads = 512; grid = plan.nteams; }
    d->last_threads = threads;
    // fewer than 8 tiles: a team inside one XCD would be bound by that XCD's share of the bandwidth (1.1 TB/s);
    // dealt over all XCDs (scatter mode: team = block / G) its sweeps run 1.4x faster and the barriers (now with
    // the L2 write-back) twice as long -- 6.1 -> 5.4 ms for 64 syndromes of the C3 code, all 50 iterations.
    // (The environment switch is read per call: a test turns it on and off.)
    const bool team_scatter = team > 1 && (exp_env("LDPC_TEAM_SCATTER") || plan.scatter);
    // scatter mode launches exactly the members (8 * team blocks for <= 4 tiles would be up to 512 workgroups,
    // more than the wide-degree instantiations can keep resident: one 8-wave workgroup per CU)
    const int team_grid = team > 1 ? (team_scatter ? plan.nteams * team : plan.grid) : 0;
    d->last_grid = team > 1 ? plan.nteams * team : grid;   // (teams: the workgroups that take part; on an XCD that hosts no team the blocks leave at once)
    d->last_kernel = team > 1 ? 4 : 1;
    d->last_team = team;
    d->last_lds_rows = 0; d->last_rows_on_chip = 0;
    const size_t slot_stride_bytes = std::max<size_t>((size_t)d->nnz, 1) * kTile * sizeof(double) + slot_pad_bytes();
    if ((st = ensure_workspace(d, (size_t)grid * slot_stride_bytes, grid, slot_stride_bytes, stream)) != LDPC_OK)
        return st;

    // Straggler hand-off in LEVELS: a tile whose active lanes have dwindled to <= T0 gives those syndromes up
    // together with their message columns (bp_kernels.hpp: packed tiles of level 1); a pass over level 1 resumes
    // them where they stood, densely packed again, and may itself hand its stragglers (<= T1 lanes) on to level 2,
    // whose pass runs them to the end.  Nothing is decoded twice, results are those of an undisturbed run
    // (the decoder is deterministic per syndrome).  Everything stays on the stream: every pass reads its syndrome
    // count from device memory, grids are sized for the level's capacity and blocks past the count return at once.
    // A level that is full refuses further tiles (they carry on by themselves), so capacities are a memory
    // budget, not a correctness bound.
    const int T0 = (d->defer_thresh < 0 || d->max_iters < 4 || ntiles < 2) ? 0
                   : (d->defer_thresh ? d->defer_thresh : d->defer_t0);
    struct Level {
        int thresh_in = 0;        // lanes at which the level below hands off into this one
        int cap_tiles = 0;        // packed tiles it can hold
        unsigned int *count = nullptr;   // device word: syndromes handed in
        unsigned int *tile_queue = nullptr, *node_queue = nullptr;
        unsigned node_take = 0;   // up to this many syndromes: the node-parallel kernel finishes them
        unsigned team_cap = 0;    // above node_take up to this many: teams of workgroups on the packed tiles
        int t_per_xcd = 0, t_gcap = 0, t_G = 0, t_nteams = 0, t_grid = 0, t_xcds = 8;   // teams on the packed tiles: members, teams
    } lv[3];
    int nlevels = 0;
    if (T0 > 0) {
        lv[1].thresh_in = T0;
        const int worst1 = (int)(((int64_t)ntiles * T0 + kTile - 1) / kTile) + 1;
        lv[1].cap_tiles = d->lvl_cap_force > 0 ? d->lvl_cap_force : std::min(worst1, std::max(8, std::max(grid, tile_grid) / 3));
        nlevels = 1;
        const int T1 = d->defer_t1;
        if (T1 > 0 && (lv[1].cap_tiles >= 8 || d->lvl_cap_force > 0)) {
            lv[2].thresh_in = T1;
            const int worst2 = (int)(((int64_t)lv[1].cap_tiles * T1 + kTile - 1) / kTile) + 1;
            lv[2].cap_tiles = d->lvl_cap_force > 0 ? d->lvl_cap_force : std::min(worst2, std::max(4, lv[1].cap_tiles / 4));
            nlevels = 2;
        }
    }
    lv[1].count = (unsigned int *)(ctrl + 4);   lv[1].tile_queue = (unsigned int *)(ctrl + 40); lv[1].node_queue = (unsigned int *)(ctrl + 48);
    lv[2].count = (unsigned int *)(ctrl + 56);  lv[2].tile_queue = (unsigned int *)(ctrl + 44); lv[2].node_queue = (unsigned int *)(ctrl + 52);
    for (int l = 1; l <= nlevels; ++l) {
        Level &L = lv[l];
        const size_t ct = (size_t)L.cap_tiles;
        if ((st = big_alloc(d->lvl_state[l - 1], ct * slot_stride_bytes, d->device)) == LDPC_ERR_UNSUPPORTED)
            st = d->lvl_state[l - 1].ensure(ct * slot_stride_bytes);
        if (st != LDPC_OK) return st;
        if ((st = d->lvl_list[l - 1].ensure(ct * kTile * sizeof(int))) != LDPC_OK) return st;
        if ((st = d->lvl_it[l - 1].ensure(ct * kTile * sizeof(int))) != LDPC_OK) return st;
        if ((st = d->lvl_syn[l - 1].ensure(std::max<size_t>(ct * s, 1) * sizeof(u64))) != LDPC_OK) return st;
        if ((st = d->lvl_never[l - 1].ensure(ct * sizeof(u64))) != LDPC_OK) return st;
        if ((st = d->lvl_err[l - 1].ensure(std::max<size_t>(ct * n, 1) * sizeof(u64))) != LDPC_OK) return st;
        if ((st = d->lvl_fin[l - 1].ensure(std::max<size_t>(ct * n, 1) * sizeof(u64))) != LDPC_OK) return st;
        if (want_llr && (st = d->lvl_llr[l - 1].ensure(std::max<size_t>(ct * n, 1) * kTile * sizeof(double))) != LDPC_OK) return st;
        HIP_TRY(hipMemsetAsync(d->lvl_never[l - 1].p, 0, ct * sizeof(u64), stream));
        // Few stragglers (the usual case): the node-parallel kernel finishes them, one workgroup per syndrome
        // straight from / into the caller's arrays, instead of a handful of tiles that each sweep the whole
        // graph with a few lanes alive.  Decided on the device: the launches of the paths not taken find the
        // level's count on the wrong side of node_take / team_cap and return at once.
        L.node_take = (d->node_ok && d->variant == 0 && d->node_take_max > 0)
                          ? (unsigned)std::min<int64_t>(d->node_take_max, (int64_t)L.cap_tiles * kTile) : 0u;
        if (L.node_take && d->max_iters <= 4096 && team_geometry(d, want_llr, &L.t_per_xcd, &L.t_gcap) && L.t_gcap >= 3) {
            // persistent teams inside the packed tiles, as many as keep the tiles in flight inside the cache budget
            // (team_fit()); they take whatever the level holds.  Nothing fits (LDPC_TEAM_CACHE_MIB=0, large graphs):
            // one tile per team, up to as many tiles as leave every team 3 members.
            int tiles_max, x = 8, t = 1, g = 3;
            if (team_fit(team_plan_in(d, L.t_per_xcd, L.t_gcap), L.cap_tiles, false, &x, &t, &g)) {
                tiles_max = L.cap_tiles;
            } else {
                x = 8; t = L.t_per_xcd / 3; g = 3;           // (the members of surplus teams are idle: round 1's geometry in fixed form)
                tiles_max = std::min(x * t, L.cap_tiles);
            }
            L.t_G = g; L.t_xcds = x; L.t_nteams = x * t; L.t_grid = 8 * g * t;
            if ((unsigned)tiles_max * kTile > L.node_take) L.team_cap = (unsigned)tiles_max * kTile;
        }
    }

    HIP_TRY(hipMemsetAsync(ctrl, 0, 64, stream));
    d->ctrl_clean[slot] = false;
    HIP_TRY(hipMemsetAsync(d->nevermask.p, 0, (size_t)ntiles * sizeof(u64), stream));
    HIP_TRY(hipEventRecord(ev[0], stream));
    // four checks (bits) per lane in the pack / unpack kernels where the caller's arrays allow 4-byte accesses
    const bool syn_v4 = s % 4 == 0 && ((uintptr_t)d_syn & 3u) == 0, err_v4 = n % 4 == 0 && ((uintptr_t)d_err & 3u) == 0;
    if (s > 0) {
        dim3 g((unsigned)((s + (syn_v4 ? 255 : 63)) / (syn_v4 ? 256 : 64)), (unsigned)ntiles);
        hipLaunchKernelGGL(syn_v4 ? pack_syndromes_kernel<4> : pack_syndromes_kernel<1>, g, dim3(64), 0, stream, d_syn, (long long)batch, (int)s,
                           (u64 *)d->synmask.p, (u64 *)d->nevermask.p, (const int *)nullptr,
                           (const unsigned int *)nullptr, 0u);
        HIP_TRY(hipGetLastError());
    }
    const int *a_row = (const int *)d->row_ptr.p, *a_eb = (const int *)d->edge_bit.p, *a_col = (const int *)d->col_ptr.p,
              *a_c2r = (const int *)d->csc2csr.p;
    const unsigned ticket = (unsigned)(d->ncalls & 0x7fffffffu) ? (unsigned)(d->ncalls & 0x7fffffffu) : 0x7fffffffu;
    // per pass: the hot parameters travel by value (BPParams), the rest in a BPCold block in device memory
    BPCold cold[3] = {};
    auto hand_off_into = [&](BPParams &q, BPCold &c, int l) {   // the pass (q, c) hands its stragglers to level l (0 = to nobody)
        if (l < 1 || l > nlevels) { q.defer_thresh = 0; c.defer_list = nullptr; c.defer_it = nullptr; c.defer_count = nullptr;
                                    c.next_state = nullptr; c.next_stride = 0; c.next_cap = 0; return; }
        q.defer_thresh = lv[l].thresh_in;
        c.defer_list = (int *)d->lvl_list[l - 1].p;
        c.defer_it = (int *)d->lvl_it[l - 1].p;
        c.defer_count = lv[l].count;
        c.next_state = (double *)d->lvl_state[l - 1].p;
        c.next_stride = (long long)(slot_stride_bytes / sizeof(double));
        c.next_cap = (unsigned)lv[l].cap_tiles * kTile;
    };
    if ((st = d->cold.ensure(3 * sizeof(BPCold))) != LDPC_OK) return st;
    BPCold *const d_cold = (BPCold *)d->cold.p;
    BPParams p{};
    p.s = (int)s; p.n = (int)n; p.nnz = (int)d->nnz; p.max_iters = (int)d->max_iters;
    p.ntiles = ntiles; p.batch = batch;
    p.r = d->per / (1 - d->per);  // belief_propagation.jl:129,153 (IEEE double division, same on host)
    p.msg = (double *)d->msg.p;
    p.slot_stride = (long long)(slot_stride_bytes / sizeof(double));
    p.errmask = (u64 *)d->errmask.p;
    p.finmask = (u64 *)d->finmask.p;
    p.llr = want_llr ? (double *)d->llr_t.p : nullptr;
    p.queue = (unsigned int *)ctrl;
    p.cold = d_cold + 0;
    p.defer_min_iter = 2;
    p.count_dev = nullptr;
    p.count_skip = 0;
    p.resumed = 0;
    p.llr_exact = d->llr_exact ? 1 : 0;
    for (BPCold &c : cold) {
        c.iters = d_iters;
        c.conv = d_conv;
        c.sum_iters = (u64 *)(ctrl + 8);
        c.phase_ticks = (u64 *)(ctrl + 16);
    }
    hand_off_into(p, cold[0], 1);
    for (int l = 1; l <= nlevels; ++l) {
        BPParams unused{};
        cold[l].index = (const int *)d->lvl_list[l - 1].p;
        cold[l].it0 = (const int *)d->lvl_it[l - 1].p;
        hand_off_into(unused, cold[l], l + 1);
    }
    hipLaunchKernelGGL(store_cold_kernel, dim3(1), dim3(64), 0, stream, cold[0], cold[1], cold[2], d_cold);
    HIP_TRY(hipGetLastError());
    bp_kernel_t kfn = pick_kernel(d->max_cdeg, d->max_bdeg, want_llr, threads);
    const int always_release = d->team_always_release ? 1 : 0;
    auto team_params = [&](DevBuf &wsbuf, int nteams, int tiles, TeamParams &tp) -> ldpc_status {
        const size_t ctl_bytes = ((size_t)nteams + 1) * kTeamCtlWords * sizeof(unsigned int);   // + the block of the tile queue
        const size_t mism_stride = ((size_t)d->max_iters + 31) & ~(size_t)31;
        const size_t ws_bytes = ctl_bytes + (size_t)tiles * mism_stride * sizeof(u64);
        ldpc_status r = wsbuf.ensure(ws_bytes);
        if (r != LDPC_OK) return r;
        HIP_TRY(hipMemsetAsync(wsbuf.p, 0, ws_bytes, stream));
        tp.ctl = (unsigned int *)wsbuf.p;
        tp.mism = (u64 *)((char *)wsbuf.p + ctl_bytes);
        tp.mism_stride = (int)mism_stride;
        tp.fault = d->team_fault_dev;
        tp.always_release = always_release;
        tp.nteams = nteams;
        tp.xcds = 8;
        tp.dynamic = d->team_dynamic;
        tp.pairs = d->team_pairs;
        tp.scatter = 0;
        tp.count_max = 0;
        tp.inject_fault = 0;
        tp.ticket = ticket;
        tp.rollcall_ticks = d->rollcall_ticks;
        tp.errmask_alt = nullptr;
        tp.ahead_min = 0;
        tp.ahead_from = 2;
        tp.llr_raw = 0;
        return LDPC_OK;
    };
    bool team_ran = team > 1;
    int llr_raw_out = 0;   // the fresh pass left posterior odds, not logarithms, in llr_t (TeamParams::llr_raw)
    const int *llr_posmap = nullptr;   // ... in the dealt bit order of the rows-on-chip tables: position of every bit
    HIP_TRY(hipEventRecord(ev[1], stream));
    if (team > 1) {
        TeamParams tp{};
        if ((st = team_params(d->team_ws, plan.nteams, ntiles, tp)) != LDPC_OK) return st;
        tp.G = team;
        tp.xcds = plan.xcds;
        tp.scatter = team_scatter ? 1 : 0;
        tp.inject_fault = d->inject_fault;   // (tests; the kernel only looks at it in the experiments build)
        tp.errmask_alt = (u64 *)d->errmask.p + std::max<size_t>((size_t)ntiles * n, 1);
        tp.ahead_min = d->team_ahead;
        tp.ahead_from = d->team_ahead_from;
        tp.llr_raw = want_llr ? d->team_llr_raw : 0;
        llr_raw_out = tp.llr_raw == 6 ? 4 : tp.llr_raw;
        // one round of teams over all XCDs (<= 3 tiles: a single decode!): no other tile waits for this team, so a sweep
        // ahead that turns out to be for nothing costs one sweep at the end, and the barrier saved in every iteration
        // is worth it whatever the number of active lanes (one syndrome, 50 iterations: 2.62 -> 2.31 ms)
        if (plan.scatter && !d->team_ahead_set && tp.ahead_min > 0) tp.ahead_min = 1;
        team_kernel_t tk = pick_team_kernel(d->max_cdeg, d->max_bdeg, want_llr);
        size_t team_lds = team_lds_bytes();
        const int *t_col = a_col, *t_c2r = a_c2r, *t_row = a_row;
        if (plan.irr && !team_scatter && team_irr_build(d, team) == LDPC_OK && d->irr_on_chip > 0) {
            // an irregular graph: whole checks in the LDS of their owners (bp_team_kernels.hpp, IRR)
            team_kernel_t tki = pick_team_kernel_irr(d->irr_dcb, d->max_bdeg, want_llr);
            const size_t need = (size_t)d->irr_R * kTile * sizeof(double);
            int occ_irr = 0;
            if (tki && d->prepare_kernel((const void *)tki, LDPC_TEAM_THREADS, need, &occ_irr) == LDPC_OK && occ_irr >= 1) {
                tk = tki; team_lds = need;
                d->last_lds_rows = d->irr_R; d->last_rows_on_chip = d->irr_on_chip;
                tp.rows.lds_edge = (const int *)d->irr_lds_edge.p;
                tp.rows.R = d->irr_R;
                tp.rows.reg_edge = nullptr; tp.rows.regs = 0;
                tp.rows.static_c = tp.rows.static_v = LDPC_TEAM_THREADS / 64; tp.rows.flip = 0;
                t_row = (const int *)d->irr_ctab.p;      // (this instantiation reads its tables through these three arguments)
                t_col = (const int *)d->irr_ptab.p;
                t_c2r = (const int *)d->irr_ploc.p;
                llr_posmap = (const int *)d->irr_posmap.p;
            }
        }
        if (plan.rows && team_rows_build(d, team) == LDPC_OK) {   // (also with LDPC_TEAM_SCATTER: members over all XCDs, a test)
            // rows that only one member touches live in its LDS (TeamRows)
            team_kernel_t tkr = pick_team_kernel_rows(d->rows_dc, d->rows_dv, want_llr, d->rows_regs > 0);
            const size_t need = (size_t)(d->rows_R + (d->rows_regs > 0 ? 1 : 0)) * kTile * sizeof(double);   // (+ the dummy row of the register rows)
            int occ_rows = 0;
            if (tkr && d->prepare_kernel((const void *)tkr, LDPC_TEAM_THREADS, need, &occ_rows) == LDPC_OK && occ_rows >= 1) {
                tk = tkr; team_lds = need;
                d->last_lds_rows = d->rows_R; d->last_rows_on_chip = d->rows_on_chip;
                tp.rows.lds_edge = (const int *)d->rows_lds_edge.p;
                tp.rows.R = d->rows_R;
                tp.rows.reg_edge = (const int *)d->rows_reg_edge.p;
                tp.rows.regs = d->rows_regs;
                tp.rows.static_c = d->rows_static_c; tp.rows.static_v = d->rows_static_v; tp.rows.flip = d->team_flip;
                tp.rows.first_c = d->rows_first_c;
                // (wide teams -- members over all XCDs, a barrier of ~20 us: every on-chip chunk fits its shadow.  n = 65536, 16,384
                //  syndromes x 50 iterations: 905.6 / 892.3 / 880.9 ms with 0 / 2 / 4 chunks a wave; profiles/r04_wide_teams.txt)
                tp.pre = std::min((plan.wide && !d->team_pre_set) ? 8 : d->team_pre, d->rows_first_c / (LDPC_TEAM_THREADS / 64));
                t_col = (const int *)d->rows_ctab.p;     // (this instantiation reads its tables through these two arguments)
                t_c2r = (const int *)d->rows_vtab.p;
                llr_posmap = (const int *)d->rows_posmap.p;
            }
        }
        const u64 *a_syn = (const u64 *)d->synmask.p, *a_nev = (const u64 *)d->nevermask.p;
        void *args[] = {&p, &tp, &t_row, &a_eb, &t_col, &t_c2r, &a_syn, &a_nev};
        const hipError_t te = launch_team_grid(d, tk, team_grid, args, stream, team_lds);
        if (te != hipSuccess) {
            // a team grid the runtime refuses must not fail the call: the tile kernel decodes the batch (one
            // workgroup per tile, same results), and teams stay off for this decoder
            (void)hipGetLastError();
            d->team_max = 1;
            threads = tile_threads; grid = tile_grid;
            d->last_kernel = 1; d->last_team = 1; d->last_lds_rows = 0; d->last_rows_on_chip = 0; d->last_grid = grid; d->last_threads = threads;
            if ((st = ensure_workspace(d, (size_t)grid * slot_stride_bytes, grid, slot_stride_bytes, stream)) != LDPC_OK) return st;
            p.msg = (double *)d->msg.p;
            kfn = pick_kernel(d->max_cdeg, d->max_bdeg, want_llr, threads);
            team_ran = false;
            llr_raw_out = 0;
            llr_posmap = nullptr;
        }
        if (team_ran && exp_env("LDPC_TEAM_DEBUG")) {   // diagnostics: which XCDs did the teams land on?
            const int nt = std::min(plan.nteams, ntiles);
            std::vector<unsigned> xm((size_t)nt);
            (void)ldpc_detail::wait_stream(stream, d->device, "LDPC_TEAM_DEBUG (stream synchronise)");
            for (int t = 0; t < nt; ++t)
                (void)hipMemcpy(&xm[(size_t)t], tp.ctl + (size_t)t * kTeamCtlWords + 32, sizeof(unsigned), hipMemcpyDeviceToHost);
            int single = 0;
            for (unsigned v : xm) single += __builtin_popcount(v) == 1;
            std::fprintf(stderr, "[ldpc] team kernel: %d tiles, %d teams x %d workgroups, grid %d; %d teams on one XCD (first masks %x %x %x)\n",
                         ntiles, plan.nteams, team, team_grid, single, xm[0], nt > 1 ? xm[1] : 0u, nt > 2 ? xm[2] : 0u);
            // per team: its members' own check / variable sweep times (100 MHz ticks over the whole call: mean, fastest,
            // slowest, and where the slowest ran: XCC_ID is not stored, HW_ID gives CU (bits 8-11), SH (12), SE (13-15))
            for (int t = 0; t < nt; ++t) {
                std::vector<unsigned> ml((size_t)team * 32);
                (void)hipMemcpy(ml.data(), tp.ctl + (size_t)t * kTeamCtlWords + kTeamCtlMember, ml.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
                double sc = 0, sv = 0; unsigned cmin = ~0u, cmax = 0, vmin = ~0u, vmax = 0; int cm = 0, vm = 0;
                for (int m = 0; m < team; ++m) {
                    const unsigned c = ml[(size_t)m * 32 + 1], v = ml[(size_t)m * 32 + 2];
                    sc += c; sv += v;
                    if (c < cmin) cmin = c;
                    if (c > cmax) { cmax = c; cm = m; }
                    if (v < vmin) vmin = v;
                    if (v > vmax) { vmax = v; vm = m; }
                }
                const unsigned hc = ml[(size_t)cm * 32 + 3], hv = ml[(size_t)vm * 32 + 3];
                std::fprintf(stderr, "[ldpc]   team %d (xcc mask %x): check mean %.0f min %u max %u (member %d cu %u se %u)  var mean %.0f min %u max %u (member %d cu %u se %u)\n",
                             t, xm[(size_t)t], sc / team, cmin, cmax, cm, (hc >> 8) & 15u, (hc >> 13) & 7u, sv / team, vmin, vmax, vm, (hv >> 8) & 15u, (hv >> 13) & 7u);
            }
        }
    }
    if (!team_ran) {
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3((unsigned)threads), 0, stream, p, a_row, a_eb, a_col, a_c2r,
                           (const u64 *)d->synmask.p, (const u64 *)d->nevermask.p);
        HIP_TRY(hipGetLastError());
    }
    // ---- passes over the packed levels (tile kernel in place / teams on packed tiles; the node kernel comes last)
    bp_kernel_t kfn2 = pick_kernel(d->max_cdeg, d->max_bdeg, want_llr, threads, true);
    int occ2 = 1;
    if (nlevels && (st = d->prepare_kernel((const void *)kfn2, threads, 0, &occ2)) != LDPC_OK) return st;
    for (int l = 1; l <= nlevels; ++l) {
        Level &L = lv[l];
        const u64 *l_syn = (const u64 *)d->lvl_syn[l - 1].p, *l_nev = (const u64 *)d->lvl_never[l - 1].p;
        if (s > 0) {
            dim3 g((unsigned)((s + (syn_v4 ? 255 : 63)) / (syn_v4 ? 256 : 64)), (unsigned)L.cap_tiles);
            hipLaunchKernelGGL(syn_v4 ? pack_syndromes_kernel<4> : pack_syndromes_kernel<1>, g, dim3(64), 0, stream, d_syn, (long long)0, (int)s,
                               (u64 *)d->lvl_syn[l - 1].p, (u64 *)d->lvl_never[l - 1].p, (const int *)d->lvl_list[l - 1].p,
                               (const unsigned int *)L.count, L.node_take);
            HIP_TRY(hipGetLastError());
        }
        BPParams q = p;
        q.msg = (double *)d->lvl_state[l - 1].p;
        q.errmask = (u64 *)d->lvl_err[l - 1].p;
        q.finmask = (u64 *)d->lvl_fin[l - 1].p;
        q.llr = want_llr ? (double *)d->lvl_llr[l - 1].p : nullptr;
        q.queue = L.tile_queue;
        q.cold = d_cold + l;
        q.resumed = 1;
        q.count_dev = L.count;
        q.count_skip = std::max(L.node_take, L.team_cap);
        q.defer_min_iter = 1;
        hand_off_into(q, cold[l], l + 1);   // (sets q.defer_thresh too)
        const int g2 = std::max(1, std::min(occ2 * d->num_cus, L.cap_tiles));
        hipLaunchKernelGGL(kfn2, dim3((unsigned)g2), dim3((unsigned)threads), 0, stream, q, a_row, a_eb, a_col, a_c2r, l_syn, l_nev);
        HIP_TRY(hipGetLastError());
        if (L.team_cap) {
            BPParams q3 = q;
            q3.count_skip = L.node_take;
            TeamParams tp{};
            if ((st = team_params(d->team_ws_lvl[l - 1], L.t_nteams, (int)(L.team_cap / kTile), tp)) != LDPC_OK) return st;
            tp.G = L.t_G;
            tp.xcds = L.t_xcds;
            tp.count_max = L.team_cap;
            team_kernel_t tk = pick_team_kernel(d->max_cdeg, d->max_bdeg, want_llr, true);
            void *args[] = {&q3, &tp, &a_row, &a_eb, &a_col, &a_c2r, &l_syn, &l_nev};
            HIP_TRY(launch_team_grid(d, tk, L.t_grid, args, stream, team_lds_bytes()));
        }
    }
    HIP_TRY(hipEventRecord(ev[2], stream));
    // ---- results out: level 0 first, then every level over the rows its lower levels gave up
    if (n > 0) {
        dim3 g((unsigned)((n + 63) / 64), (unsigned)ntiles);
        const dim3 ge((unsigned)((n + (err_v4 ? 255 : 63)) / (err_v4 ? 256 : 64)), (unsigned)ntiles);
        hipLaunchKernelGGL(err_v4 ? unpack_errors_kernel<4> : unpack_errors_kernel<1>, ge, dim3(64), 0, stream, (const u64 *)d->finmask.p,
                           (long long)batch, (int)n, d_err, (const int *)nullptr, (const unsigned int *)nullptr, 0u);
        HIP_TRY(hipGetLastError());
        if (want_llr) {
            const dim3 gl(g.x, (g.y + 7u) & ~7u);     // (unpack_llr_kernel deals the tiles over the XCDs in eights)
            hipLaunchKernelGGL(unpack_llr_kernel, gl, dim3(256), 0, stream, (const double *)d->llr_t.p,
                               (long long)batch, (int)n, d_llr, (const int *)nullptr, (const unsigned int *)nullptr, 0u, llr_raw_out, d->llr_exact ? 1 : 0, llr_posmap);
            HIP_TRY(hipGetLastError());
        }
        for (int l = 1; l <= nlevels; ++l) {
            dim3 g2((unsigned)((n + 63) / 64), (unsigned)lv[l].cap_tiles);
            const dim3 ge2((unsigned)((n + (err_v4 ? 255 : 63)) / (err_v4 ? 256 : 64)), (unsigned)lv[l].cap_tiles);
            hipLaunchKernelGGL(err_v4 ? unpack_errors_kernel<4> : unpack_errors_kernel<1>, ge2, dim3(64), 0, stream, (const u64 *)d->lvl_fin[l - 1].p,
                               (long long)0, (int)n, d_err, (const int *)d->lvl_list[l - 1].p,
                               (const unsigned int *)lv[l].count, lv[l].node_take);
            HIP_TRY(hipGetLastError());
            if (want_llr) {
                const dim3 gl2(g2.x, (g2.y + 7u) & ~7u);
                hipLaunchKernelGGL(unpack_llr_kernel, gl2, dim3(256), 0, stream, (const double *)d->lvl_llr[l - 1].p,
                                   (long long)0, (int)n, d_llr, (const int *)d->lvl_list[l - 1].p,
                                   (const unsigned int *)lv[l].count, lv[l].node_take, 0, 0, (const int *)nullptr);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    // ---- levels that hold only a few syndromes: the node-parallel kernel resumes each from its packed column
    for (int l = 1; l <= nlevels; ++l) {
        Level &L = lv[l];
        if (!L.node_take) continue;
        const bool mlds = d->node_msg_lds;
        const int msg_mode = mlds ? 1 : (d->node_split_check > 0 ? 2 : 0);
        const int nthreads = msg_mode ? 1024 : 512;   // (global slots: 1024 x 1 per CU measured the same as 512 x 2)
        const size_t nlds = node_lds_bytes((int)s, (int)n) +
                            (msg_mode == 1 ? (size_t)d->nnz : msg_mode == 2 ? (size_t)d->node_split_edge : 0) * sizeof(double);
        node_kernel_t nk = pick_node_kernel(d->max_cdeg, d->max_bdeg, want_llr, nthreads, msg_mode);
        int per_cu_unused = 0;
        if ((st = d->prepare_kernel((const void *)nk, nthreads, nlds, &per_cu_unused)) != LDPC_OK) return st;
        const int ngrid = (int)std::min<int64_t>((int64_t)L.node_take, (int64_t)(msg_mode ? 1 : 2) * d->num_cus);
        const size_t stride = (std::max<size_t>((size_t)d->nnz, 1) + 63) & ~(size_t)63;
        if ((st = d->node_msg.ensure(mlds ? 64 : (size_t)ngrid * stride * sizeof(double))) != LDPC_OK) return st;
        NodeParams np{};
        np.s = (int)s; np.n = (int)n; np.nnz = (int)d->nnz; np.max_iters = (int)d->max_iters;
        np.batch = 0; np.r = p.r;
        np.syn = d_syn; np.err = d_err; np.conv = d_conv; np.iters = d_iters; np.llr = d_llr; np.llr_exact = d->llr_exact ? 1 : 0;
        np.msg = (double *)d->node_msg.p; np.slot_stride = (long long)stride;
        np.queue = L.node_queue;
        np.sum_iters = (u64 *)(ctrl + 8);
        np.index = (const int *)d->lvl_list[l - 1].p; np.count_dev = L.count; np.count_max = L.node_take;
        np.it0 = (const int *)d->lvl_it[l - 1].p;
        np.state = (const double *)d->lvl_state[l - 1].p;
        np.state_stride = (long long)(slot_stride_bytes / sizeof(double));
        np.done_count = nullptr; np.done_flag = nullptr; np.done_ticket = 0; np.next_ctrl = nullptr;
        np.split_check = d->node_split_check; np.split_edge = d->node_split_edge;
        hipLaunchKernelGGL(nk, dim3((unsigned)ngrid), dim3((unsigned)nthreads), nlds, stream, np, a_row, a_eb, a_col, a_c2r);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ev[3], stream));
    d->timed[slot] = true;
    d->last_ev = ev[3];
    return LDPC_OK;
}

extern "C" {

ldpc_status ldpc_bp_decode_batch_device(ldpc_bp_decoder *d, int64_t batch, const uint8_t *d_syn,
                                        uint8_t *d_err, uint8_t *d_conv, double *d_llr, int32_t *d_iters,
                                        void *stream_v)
{
    return decode_device_impl(d, batch, d_syn, d_err, d_conv, d_llr, d_iters, stream_v, nullptr);
}

static ldpc_status decode_batch_host_impl(ldpc_bp_decoder *d, int64_t batch, const uint8_t *syn, uint8_t *err,
                                          uint8_t *conv, double *llr, int32_t *iters);

ldpc_status ldpc_bp_decode_batch(ldpc_bp_decoder *d, int64_t batch, const uint8_t *syn, uint8_t *err,
                                 uint8_t *conv, double *llr, int32_t *iters)
{
    ldpc_status st = decode_batch_host_impl(d, batch, syn, err, conv, llr, iters);
    // This entry is synchronous, so a team that lost a workgroup (bp_team_kernels.hpp: bounded polls, fault
    // word) is known by now: do not hand the caller garbage -- decode once more without teams, and keep them
    // off for this decoder.  (The device entry cannot know yet; there the NEXT call reports it.)
    if (st == LDPC_OK && d && report_team_fault(d) != LDPC_OK)   // (clears the word, turns teams off)
        st = decode_batch_host_impl(d, batch, syn, err, conv, llr, iters);
    return st;
}

static ldpc_status decode_batch_host_impl(ldpc_bp_decoder *d, int64_t batch, const uint8_t *syn, uint8_t *err,
                                          uint8_t *conv, double *llr, int32_t *iters)
{
    if (!d) return fail(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return fail(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return ldpc_bp_decode_batch_device(d, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    if ((d->s > 0 && !syn) || (d->n > 0 && !err) || !conv)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "syndromes/errors/converged pointer is NULL");
    HIP_TRY(hipSetDevice(d->device));
    const size_t s = (size_t)d->s, n = (size_t)d->n, B = (size_t)batch;
    ldpc_status st;
    hipStream_t stream = nullptr;
    // Small batches (decode! is batch = 1): one pinned staging image, ONE copy in and ONE copy out
    // instead of five pageable transfers -- the call is latency-bound, not bandwidth-bound.
    {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_err = up(B * s), o_conv = o_err + up(B * n), o_it = o_conv + up(B),
                     o_llr = o_it + up(B * sizeof(int32_t)), total = o_llr + (llr ? up(B * n * sizeof(double)) : 0);
        // Tiny batches on the two kernels that read their input once (a plain decode! above all): the
        // kernel works on a host-mapped image directly and raises a flag in it when the last workgroup
        // is through -- ONE runtime call (the launch) instead of nine, no copies, no stream synchronisation.
        static const bool lat_off = exp_env("LDPC_NO_LATENCY_PATH") != nullptr;
        const bool lds_k = takes_lds_kernel(d, llr != nullptr);
        const int64_t lat_groups = lds_k ? ((batch + (1ll << d->lds_logS[llr ? 1 : 0]) - 1) >> d->lds_logS[llr ? 1 : 0]) : batch;
        if (!lat_off && total <= ((size_t)256 << 10) && d->max_iters > 0 && lat_groups <= 2 * (int64_t)d->num_cus &&
            (lds_k || takes_node_kernel(d, batch, llr != nullptr))) {
            const size_t hdr = 256;
            if (d->lat_pin_cap < hdr + total) {
                if (d->lat_pin) (void)hipHostFree(d->lat_pin);
                d->lat_pin = nullptr; d->lat_pin_cap = 0;
                const size_t cap = hdr + ((size_t)256 << 10);
                HIP_TRY(hipHostMalloc(&d->lat_pin, cap, hipHostMallocMapped | hipHostMallocCoherent));
                std::memset(d->lat_pin, 0, hdr);
                HIP_TRY(hipHostGetDevicePointer(&d->lat_pin_dev, d->lat_pin, 0));
                d->lat_pin_cap = cap;
            }
            char *hp = (char *)d->lat_pin + hdr, *dp = (char *)d->lat_pin_dev + hdr;
            volatile unsigned int *flag = (volatile unsigned int *)d->lat_pin;
            if (++d->lat_ticket == 0) d->lat_ticket = 1;
            const LatencyCtl lc{(unsigned int *)d->lat_pin_dev, d->lat_ticket};
            std::memcpy(hp, syn, B * s);
            st = decode_device_impl(d, batch, (const uint8_t *)dp, (uint8_t *)(dp + o_err), (uint8_t *)(dp + o_conv),
                                    llr ? (double *)(dp + o_llr) : nullptr, (int32_t *)(dp + o_it), stream, &lc);
            if (st != LDPC_OK) return st;
            const auto lat_t0 = std::chrono::steady_clock::now();
            for (uint64_t spins = 1;; ++spins) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == lc.ticket) break;
                if ((spins & 0xffff) == 0) {   // every ~65k polls: is the kernel still alive?  (and the bound of host_wait.hpp)
                    const int64_t lim = ldpc_detail::wait_limit_ms();
                    if (lim > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - lat_t0).count() > lim)
                        return ldpc_detail::wait_stream(stream, d->device, "latency path (flag of the last workgroup)");   // (expires at once: names the wait, marks the device)
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) {
                        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == lc.ticket) break;
                        return fail(LDPC_ERR_HIP, "latency path: the kernel finished without raising its flag");
                    }
                    if (q != hipErrorNotReady) {
                        (void)hipGetLastError();
                        return fail(LDPC_ERR_HIP, std::string("latency path: ") + hipGetErrorString(q));
                    }
                }
                __builtin_ia32_pause();
            }
            std::memcpy(err, hp + o_err, B * n);
            std::memcpy(conv, hp + o_conv, B);
            if (iters) std::memcpy(iters, hp + o_it, B * sizeof(int32_t));
            if (llr) std::memcpy(llr, hp + o_llr, B * n * sizeof(double));
            return LDPC_OK;
        }
        if (total <= ((size_t)4 << 20)) {
            if ((st = d->st_all.ensure(total)) != LDPC_OK) return st;
            if (d->pin_cap < total) {
                if (d->pin) (void)hipHostFree(d->pin);
                d->pin = nullptr; d->pin_cap = 0;
                HIP_TRY(hipHostMalloc(&d->pin, total, hipHostMallocDefault));
                d->pin_cap = total;
            }
            char *hp = (char *)d->pin, *dp = (char *)d->st_all.p;
            std::memcpy(hp, syn, B * s);
            if (s > 0) HIP_TRY(hipMemcpyAsync(dp, hp, B * s, hipMemcpyHostToDevice, stream));
            st = ldpc_bp_decode_batch_device(d, batch, (const uint8_t *)dp, (uint8_t *)(dp + o_err), (uint8_t *)(dp + o_conv),
                                             llr ? (double *)(dp + o_llr) : nullptr, (int32_t *)(dp + o_it), stream);
            if (st != LDPC_OK) return st;
            HIP_TRY(hipMemcpyAsync(hp + o_err, dp + o_err, total - o_err, hipMemcpyDeviceToHost, stream));
            if ((st = ldpc_detail::wait_stream(stream, d->device, "ldpc_bp_decode_batch (small batch: stream synchronise)")) != LDPC_OK) return st;
            std::memcpy(err, hp + o_err, B * n);
            std::memcpy(conv, hp + o_conv, B);
            if (iters) std::memcpy(iters, hp + o_it, B * sizeof(int32_t));
            if (llr) std::memcpy(llr, hp + o_llr, B * n * sizeof(double));
            return LDPC_OK;
        }
    }
    // Large batches: chunks flow through a 3-slot pipeline -- host memcpy into pinned memory (a few
    // threads), H2D on a copy stream, decode on the compute stream, D2H on a second copy stream, host
    // memcpy out -- so that PCIe runs at pinned-memory speed in both directions while the GPU decodes.
    // (A single pageable hipMemcpy each way measured 11 GB/s and made the call 5-7x slower than the
    // HBM-resident entry for small codes.)
    {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t bps = s + n + 1 + sizeof(int32_t) + (llr ? n * sizeof(double) : 0);   // bytes per syndrome
        const bool lds_path = d->variant != 1 && d->lds_logS[llr ? 1 : 0] >= 0;
        size_t chunk_mb = 24;
        if (const char *e = exp_env("LDPC_PIPE_CHUNK_MB")) chunk_mb = std::max<long>(1, std::atol(e));
        size_t cb = (chunk_mb << 20) / std::max<size_t>(bps, 1);
        cb = std::max<size_t>(cb, lds_path ? 32768 : 65536);      // enough syndromes to fill the chip
        cb = (cb + 4095) & ~(size_t)4095;
        if (B < cb + cb / 2) cb = B;                               // no tiny trailing chunk
        const size_t nchunks = (B + cb - 1) / cb;
        const size_t o_err = up(cb * s), o_conv = o_err + up(cb * n), o_it = o_conv + up(cb),
                     o_llr = o_it + up(cb * sizeof(int32_t)), total = o_llr + (llr ? up(cb * n * sizeof(double)) : 0);
        for (int q = 0; q < 3; ++q)
            if (!d->pipe_stream[q]) HIP_TRY(hipStreamCreateWithFlags(&d->pipe_stream[q], hipStreamNonBlocking));
        for (auto &row : d->pipe_ev)
            for (hipEvent_t &e : row)
                if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const int R = (int)std::min<size_t>(ldpc_bp_decoder::kPipe, nchunks);
        if (d->pipe_pin_cap < total) {
            for (void *&q : d->pipe_pin) {
                if (q) (void)hipHostFree(q);
                q = nullptr;
            }
            d->pipe_pin_cap = 0;
        }
        for (int q = 0; q < R; ++q) {
            if (!d->pipe_pin[q]) HIP_TRY(hipHostMalloc(&d->pipe_pin[q], total, hipHostMallocDefault));
            if ((st = d->pipe_dev[q].ensure(total)) != LDPC_OK) return st;
        }
        d->pipe_pin_cap = std::max(d->pipe_pin_cap, total);
        hipStream_t s_in = d->pipe_stream[0], s_comp = d->pipe_stream[1], s_out = d->pipe_stream[2];
        auto drain = [&](size_t j) -> ldpc_status {
            const int slot = (int)(j % (size_t)R);
            const size_t b0 = j * cb, nb = std::min(cb, B - b0);
            {
                const ldpc_status wst = ldpc_detail::wait_event(d->pipe_ev[slot][2], d->device, "ldpc_bp_decode_batch (host pipeline: results of a chunk)");
                if (wst != LDPC_OK) return wst;
            }
            const char *hp = (const char *)d->pipe_pin[slot];
            parallel_memcpy(err + b0 * n, hp + o_err, nb * n);
            std::memcpy(conv + b0, hp + o_conv, nb);
            if (iters) std::memcpy(iters + b0, hp + o_it, nb * sizeof(int32_t));
            if (llr) parallel_memcpy(llr + b0 * n, hp + o_llr, nb * n * sizeof(double));
            return LDPC_OK;
        };
        ldpc_status pst = LDPC_OK;
        for (size_t k = 0; k < nchunks && pst == LDPC_OK; ++k) {
            const int slot = (int)(k % (size_t)R);
            const size_t b0 = k * cb, nb = std::min(cb, B - b0);
            if (k >= (size_t)R && (pst = drain(k - R)) != LDPC_OK) break;
            char *hp = (char *)d->pipe_pin[slot], *dp = (char *)d->pipe_dev[slot].p;
            parallel_memcpy(hp, syn + b0 * s, nb * s);
            hipError_t e = hipSuccess;
            if (s > 0) e = hipMemcpyAsync(dp, hp, nb * s, hipMemcpyHostToDevice, s_in);
            if (e == hipSuccess) e = hipEventRecord(d->pipe_ev[slot][0], s_in);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_comp, d->pipe_ev[slot][0], 0);
            if (e != hipSuccess) { pst = fail(LDPC_ERR_HIP, std::string("host pipeline (H2D): ") + hipGetErrorString(e)); break; }
            pst = ldpc_bp_decode_batch_device(d, (int64_t)nb, (const uint8_t *)dp, (uint8_t *)(dp + o_err),
                                              (uint8_t *)(dp + o_conv), llr ? (double *)(dp + o_llr) : nullptr,
                                              (int32_t *)(dp + o_it), s_comp);
            if (pst != LDPC_OK) break;
            e = hipEventRecord(d->pipe_ev[slot][1], s_comp);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_out, d->pipe_ev[slot][1], 0);
            if (e == hipSuccess) e = hipMemcpyAsync(hp + o_err, dp + o_err, total - o_err, hipMemcpyDeviceToHost, s_out);
            if (e == hipSuccess) e = hipEventRecord(d->pipe_ev[slot][2], s_out);
            if (e != hipSuccess) { pst = fail(LDPC_ERR_HIP, std::string("host pipeline (D2H): ") + hipGetErrorString(e)); break; }
        }
        if (pst == LDPC_OK)
            for (size_t j = nchunks > (size_t)R ? nchunks - R : 0; j < nchunks && pst == LDPC_OK; ++j) pst = drain(j);
        if (pst != LDPC_OK) {
            const std::string keep = g_err;
            (void)ldpc_detail::wait_device(d->device, "ldpc_bp_decode_batch (host pipeline: drain after an error)");   // nothing of this call may still be in flight when we return
            g_err = keep;
        }
        return pst;
    }
}

ldpc_status ldpc_bp_last_status(ldpc_bp_decoder *d)
{
    if (!d) return fail(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    HIP_TRY(hipSetDevice(d->device));
    if (d->last_ev) {
        const ldpc_status wst = ldpc_detail::wait_event(d->last_ev, d->device, "ldpc_bp_last_status (wait for the handle's last call)");
        if (wst != LDPC_OK) return wst;
    }
    return report_team_fault(d);
}

ldpc_status ldpc_bp_call_timing(ldpc_bp_decoder *d, int32_t calls_back, double *sweep_ms, double *total_ms,
                                int64_t *sum_iters)
{
    if (!d) return fail(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (sweep_ms) *sweep_ms = 0.0;
    if (total_ms) *total_ms = 0.0;
    if (sum_iters) *sum_iters = 0;
    if (calls_back < 0 || calls_back >= ldpc_bp_decoder::kHistory || (uint64_t)calls_back >= d->ncalls)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "calls_back outside the timing ring");
    const int slot = (int)((d->ncalls - 1 - (uint64_t)calls_back) % ldpc_bp_decoder::kRing);
    if (!d->timed[slot]) return LDPC_OK;
    HIP_TRY(hipSetDevice(d->device));
    const int e0 = d->two_events[slot] ? 1 : 0, e3 = d->two_events[slot] ? 2 : 3;
    {
        const ldpc_status wst = ldpc_detail::wait_event(d->ev[slot][e3], d->device, "ldpc_bp_call_timing (wait for that call)");
        if (wst != LDPC_OK) return wst;
    }
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, d->ev[slot][1], d->ev[slot][2]));
    HIP_TRY(hipEventElapsedTime(&b, d->ev[slot][e0], d->ev[slot][e3]));
    if (sweep_ms) *sweep_ms = a;
    if (total_ms) *total_ms = b;
    if (sum_iters) {
        u64 v = 0;
        HIP_TRY(hipMemcpy(&v, (char *)d->ctrl.p + 64 * slot + 8, sizeof v, hipMemcpyDeviceToHost));
        *sum_iters = (int64_t)v;
    }
    return LDPC_OK;
}

ldpc_status ldpc_bp_call_phase_ticks(ldpc_bp_decoder *d, int32_t calls_back, uint64_t ticks[3])
{
    if (!d || !ticks) return fail(LDPC_ERR_INVALID_ARGUMENT, "NULL argument");
    ticks[0] = ticks[1] = ticks[2] = 0;
    if (calls_back < 0 || calls_back >= ldpc_bp_decoder::kHistory || (uint64_t)calls_back >= d->ncalls)
        return fail(LDPC_ERR_INVALID_ARGUMENT, "calls_back outside the timing ring");
    const int slot = (int)((d->ncalls - 1 - (uint64_t)calls_back) % ldpc_bp_decoder::kRing);
    if (!d->timed[slot]) return LDPC_OK;
    HIP_TRY(hipSetDevice(d->device));
    {
        const ldpc_status wst = ldpc_detail::wait_event(d->ev[slot][d->two_events[slot] ? 2 : 3], d->device, "ldpc_bp_call_phase_ticks (wait for that call)");
        if (wst != LDPC_OK) return wst;
    }
    HIP_TRY(hipMemcpy(ticks, (char *)d->ctrl.p + 64 * slot + 16, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return LDPC_OK;
}

ldpc_status ldpc_bp_last_timing(ldpc_bp_decoder *d, double *sweep_ms, double *total_ms, int64_t *sum_iters)
{
    if (d && d->ncalls == 0) {
        if (sweep_ms) *sweep_ms = 0.0;
        if (total_ms) *total_ms = 0.0;
        if (sum_iters) *sum_iters = 0;
        return LDPC_OK;
    }
    return ldpc_bp_call_timing(d, 0, sweep_ms, total_ms, sum_iters);
}

}  // extern "C"
