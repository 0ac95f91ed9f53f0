// ldpc_multi.hip -- batchdecode! partitioned over several GPUs of one node, behind the C ABI
// (include/ldpc_mi355x.h: ldpc_bp_create_multi / ldpc_bp_decode_batch_multi[_device]).
//
// Reference interface replaced: `batchdecode!(decoder, syndromes, errors, success)`
// (src/decoders/belief_propagation.jl:220-231) -- ONE caller-held s x B matrix, ONE call -- for a Julia host that
// reaches the library through `ccall` from a single process.  The columns are decoded independently (:224-228), so
// the path shards with no collective inside the decode: logical device g gets the contiguous columns
// [g*B/G, (g+1)*B/G), decodes them with its own single-device handle (ldpc_mi355x.hip: its own copy
// of the Tanner graph, its own workspace) on its own stream, and the hard decisions / flags (/ iteration counts /
// LLRs) come back into the caller's arrays.
//
//   * HOST form: every shard goes pinned host -> ITS OWN GPU -> pinned host through that handle's 3-slot pipeline
//     (ldpc_bp_decode_batch), driven by one host thread per device.  No hop through GPU 0.
//   * ROOT-DEVICE form: the batch is resident in the HBM of devices[0].  Shards travel to the peers and results back
//     with RCCL point-to-point operations from this one process: one communicator per device (ncclCommInitAll),
//     ncclGroupStart / ncclSend on the root's stream + ncclRecv on each peer's stream / ncclGroupEnd -- over xGMI each
//     peer has its own direct link to the root.  RCCL is bound at run time (dlopen of librccl.so.1 -- the copy that is
//     already in the process when the host is PyTorch) so that single-GPU users never load it.  Logical devices that
//     share a GPU (a rehearsal on fewer GPUs than shards) cannot form an RCCL clique; their exchange is
//     hipMemcpyPeerAsync, ordered by events.
//
// There is no CPU path here either: everything decodes through ldpc_bp_decode_batch[_device].
#include "../../include/ldpc_mi355x.h"
#include "host_env.hpp"
#include "host_wait.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using ldpc_detail::set_error;   // (ldpc_mi355x.hip, declared in host_wait.hpp)

namespace {

#define MHIP_TRY(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (void)hipGetLastError();                                                                \
            return set_error(e_ == hipErrorOutOfMemory ? LDPC_ERR_OUT_OF_MEMORY : LDPC_ERR_HIP,     \
                             std::string(#expr) + ": " + hipGetErrorString(e_));                    \
        }                                                                                           \
    } while (0)

// ---- RCCL, bound at run time
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;   // why it could not be bound
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(r.lib, n);
            if (!p && r.why.empty()) r.why = std::string("librccl has no symbol ") + n;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r.why.empty() ? &r : nullptr;
}

#define NCCL_TRY(R, expr)                                                                                      \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return set_error(LDPC_ERR_HIP, std::string(#expr) + ": " + (R)->GetErrorString(r_)); \
    } while (0)

// An RCCL group that has been opened is closed on every path out of the scope (the calling thread must not stay inside
// a group).  Between ncclGroupStart and ncclGroupEnd ONLY RCCL's own calls can fail here -- buffers, streams and device
// ordinals are all settled before the group opens --, and a group in which an RCCL call has failed is discarded by
// ncclGroupEnd, not launched (NCCL 2.27 group.cc: the thread's group error makes ncclGroupEnd clean the queued tasks up
// and return that error), so closing it cannot send the caller's stream waiting for a peer that never comes.  After
// such a failure the communicators are aborted (exchange_failed()): the next root-device call builds new ones.
struct GroupGuard {
    Rccl *R;
    bool open = false;
    explicit GroupGuard(Rccl *r) : R(r) {}
    ncclResult_t start() { const ncclResult_t r = R->GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return R->GroupEnd(); }
    ~GroupGuard() { if (open) (void)R->GroupEnd(); }
};

struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Buf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        else { (void)hipGetLastError(); p = nullptr; }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }   // (destroy_multi: after the device has been waited for)
};

}  // namespace

struct ldpc_bp_multi {
    int ndev = 0;
    int64_t s = 0, n = 0;
    int exchange = 0;                       // LDPC_EXCHANGE_*: what the root-device form moves shards with
    std::vector<int> dev;                   // HIP ordinal of logical device g (g = 0 is the root)
    std::vector<ldpc_bp_decoder *> h;       // one single-device handle each
    std::vector<hipStream_t> st;            // peers' streams (st[0]: only for the self-exchange rehearsal)
    struct Shard { Buf syn, err, conv, iters, llr; };
    std::vector<Shard> shard;               // peers' shard buffers in THEIR HBM (grow only)
    std::vector<hipEvent_t> ev_done;        // per device: its shard is decoded (copy exchange)
    hipEvent_t ev_in = nullptr;             // the caller's batch is ready on the root stream (copy exchange)
    hipEvent_t ev_t[4] = {};                // root stream: call begins / scatter enqueued / root shard decoded / gather done
    std::vector<ncclComm_t> comm;           // RCCL exchange: one communicator per device, created on first use
    bool comm_ready = false;
    bool timed = false;
    int64_t last_batch = 0;
    bool last_llr = false, last_iters = false;
};

namespace {

void shard_bounds(int64_t batch, int G, int g, int64_t *lo, int64_t *hi)
{
    *lo = batch * g / G;
    *hi = batch * (g + 1) / G;
}

void destroy_multi(ldpc_bp_multi *m)
{
    if (!m) return;
    DeviceGuard guard;
    for (int g = 0; g < (int)m->h.size(); ++g)
        if (m->h[(size_t)g]) (void)ldpc_bp_destroy(m->h[(size_t)g]);   // (synchronises its device)
    if (m->comm_ready) {
        Rccl *R = rccl();
        for (ncclComm_t c : m->comm)
            if (R && c) (void)R->CommDestroy(c);
    }
    for (int g = 0; g < (int)m->dev.size(); ++g) {
        if (ldpc_detail::device_stalled(m->dev[(size_t)g])) continue;   // (host_wait.hpp: nothing of a stalled device is freed)
        (void)hipSetDevice(m->dev[(size_t)g]);
        if (g < (int)m->shard.size()) {
            ldpc_bp_multi::Shard &sh = m->shard[(size_t)g];
            for (Buf *b : {&sh.syn, &sh.err, &sh.conv, &sh.iters, &sh.llr}) b->release();
        }
        if (g < (int)m->st.size() && m->st[(size_t)g]) (void)hipStreamDestroy(m->st[(size_t)g]);
        if (g < (int)m->ev_done.size() && m->ev_done[(size_t)g]) (void)hipEventDestroy(m->ev_done[(size_t)g]);
    }
    if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
    if (m->ev_in) (void)hipEventDestroy(m->ev_in);
    for (hipEvent_t &e : m->ev_t)
        if (e) (void)hipEventDestroy(e);
    delete m;
}

// Everything enqueued by a root-device call that is being given up -- on the peers' streams and on the root's -- has
// finished (or the bounded wait has named what did not) before the call returns: the caller may free or reuse its
// arrays, and the next call finds the shard buffers idle.
void drain_streams(ldpc_bp_multi *m, hipStream_t root)
{
    const std::string keep = ldpc_last_error();
    for (int g = 1; g < m->ndev; ++g)
        if (m->st[(size_t)g] && hipSetDevice(m->dev[(size_t)g]) == hipSuccess)
            (void)ldpc_detail::wait_stream(m->st[(size_t)g], m->dev[(size_t)g], "multi-device call given up: drain of a peer's stream");
    if (hipSetDevice(m->dev[0]) == hipSuccess)
        (void)ldpc_detail::wait_stream(root, m->dev[0], "multi-device call given up: drain of the root's stream");
    (void)hipGetLastError();
    (void)set_error(LDPC_ERR_HIP, keep);   // (the message of the failure itself, not of the drain)
}

// An RCCL call failed inside a group (the group has been closed and discarded by then, see GroupGuard): the
// communicators are in an unknown state -- abort them; the next root-device call creates new ones.
void exchange_failed(ldpc_bp_multi *m, hipStream_t root)
{
    drain_streams(m, root);
    const std::string keep = ldpc_last_error();
    if (m->comm_ready) {
        Rccl *R = rccl();
        for (ncclComm_t &c : m->comm) {
            if (R && c) (void)(R->CommAbort ? R->CommAbort(c) : R->CommDestroy(c));
            c = nullptr;
        }
        m->comm_ready = false;
    }
    (void)set_error(LDPC_ERR_HIP, keep);
}

ldpc_status ensure_comms(ldpc_bp_multi *m)
{
    if (m->comm_ready) return LDPC_OK;
    Rccl *R = rccl();
    if (!R) return set_error(LDPC_ERR_UNSUPPORTED, "RCCL is not available: the root-device form needs librccl.so.1 for more than one GPU");
    m->comm.assign((size_t)m->ndev, nullptr);
    NCCL_TRY(R, R->CommInitAll(m->comm.data(), m->ndev, m->dev.data()));
    m->comm_ready = true;
    return LDPC_OK;
}

}  // namespace

extern "C" {

ldpc_status ldpc_bp_create_multi(int32_t ndev, const int32_t *devices, int32_t exchange, int64_t s, int64_t n, int64_t nnz,
                                 const int64_t *colptr, const int64_t *rowval, double per, int64_t max_iters,
                                 const ldpc_bp_options *options, ldpc_bp_multi **out)
{
    if (!out) return set_error(LDPC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (ndev < 1 || ndev > LDPC_MULTI_MAX_DEVICES || !devices) return set_error(LDPC_ERR_INVALID_ARGUMENT, "ndev must be 1 ... 16 and devices non-NULL");
    if (exchange < LDPC_EXCHANGE_AUTO || exchange > LDPC_EXCHANGE_RCCL) return set_error(LDPC_ERR_INVALID_ARGUMENT, "exchange must be LDPC_EXCHANGE_AUTO, _COPY or _RCCL");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return set_error(LDPC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    bool distinct = true;
    for (int g = 0; g < ndev; ++g) {
        if (devices[g] < 0 || devices[g] >= count) return set_error(LDPC_ERR_INVALID_ARGUMENT, "device ordinal out of range");
        for (int q = 0; q < g; ++q) distinct = distinct && devices[q] != devices[g];
    }
    if (exchange == LDPC_EXCHANGE_RCCL && !distinct)
        return set_error(LDPC_ERR_INVALID_ARGUMENT, "logical devices that share a GPU cannot form an RCCL clique: use LDPC_EXCHANGE_AUTO or _COPY");
    ldpc_bp_multi *m = new (std::nothrow) ldpc_bp_multi();
    if (!m) return set_error(LDPC_ERR_OUT_OF_MEMORY, "host allocation failed");
    m->ndev = ndev; m->s = s; m->n = n;
    m->dev.assign(devices, devices + ndev);
    // one GPU: nothing to exchange (unless RCCL is asked for by name: the shard then travels to itself through a one-rank
    // communicator -- the rehearsal of the RCCL calls on a one-GPU box)
    m->exchange = ndev == 1 ? (exchange == LDPC_EXCHANGE_RCCL ? LDPC_EXCHANGE_RCCL : LDPC_EXCHANGE_NONE)
                            : (exchange == LDPC_EXCHANGE_AUTO ? (distinct ? LDPC_EXCHANGE_RCCL : LDPC_EXCHANGE_COPY) : exchange);
    m->h.assign((size_t)ndev, nullptr);
    m->st.assign((size_t)ndev, nullptr);
    m->shard.resize((size_t)ndev);
    m->ev_done.assign((size_t)ndev, nullptr);
    DeviceGuard guard;
    for (int g = 0; g < ndev; ++g) {
        ldpc_bp_options o;
        if (options) o = *options; else std::memset(&o, 0, sizeof o);
        o.device = devices[g];
        const ldpc_status st = ldpc_bp_create(s, n, nnz, colptr, rowval, per, max_iters, &o, &m->h[(size_t)g]);
        if (st != LDPC_OK) { destroy_multi(m); return st; }
        if (hipSetDevice(devices[g]) != hipSuccess ||
            hipStreamCreateWithFlags(&m->st[(size_t)g], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->ev_done[(size_t)g], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            destroy_multi(m);
            return set_error(LDPC_ERR_HIP, "stream / event creation failed");
        }
    }
    if (hipSetDevice(devices[0]) != hipSuccess || hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        destroy_multi(m);
        return set_error(LDPC_ERR_HIP, "event creation failed");
    }
    for (hipEvent_t &e : m->ev_t)
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); destroy_multi(m); return set_error(LDPC_ERR_HIP, "event creation failed"); }
    *out = m;
    return LDPC_OK;
}

ldpc_status ldpc_bp_destroy_multi(ldpc_bp_multi *m)
{
    destroy_multi(m);
    return LDPC_OK;
}

ldpc_bp_decoder *ldpc_bp_multi_handle(ldpc_bp_multi *m, int32_t g)
{
    if (!m || g < 0 || g >= m->ndev) return nullptr;
    return m->h[(size_t)g];
}

// ---- HOST form: one host thread per logical device, each shard through its own device's pipeline
ldpc_status ldpc_bp_decode_batch_multi(ldpc_bp_multi *m, int64_t batch, const uint8_t *syn, uint8_t *err, uint8_t *conv,
                                       double *llr, int32_t *iters)
{
    if (!m) return set_error(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return set_error(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return LDPC_OK;
    if ((m->s > 0 && !syn) || (m->n > 0 && !err) || !conv) return set_error(LDPC_ERR_INVALID_ARGUMENT, "syndromes/errors/converged pointer is NULL");
    m->timed = false;
    const int G = m->ndev;
    std::vector<ldpc_status> st((size_t)G, LDPC_OK);
    std::vector<std::string> msg((size_t)G);
    auto work = [&](int g) {
        int64_t lo, hi;
        shard_bounds(batch, G, g, &lo, &hi);
        if (hi <= lo) return;
        (void)hipSetDevice(m->dev[(size_t)g]);   // (the current device is a per-thread setting)
        st[(size_t)g] = ldpc_bp_decode_batch(m->h[(size_t)g], hi - lo, syn + (size_t)lo * (size_t)m->s, err + (size_t)lo * (size_t)m->n,
                                             conv + lo, llr ? llr + (size_t)lo * (size_t)m->n : nullptr, iters ? iters + lo : nullptr);
        if (st[(size_t)g] != LDPC_OK) msg[(size_t)g] = ldpc_last_error();   // (thread-local: carried over to the caller below)
    };
    if (G == 1) {
        DeviceGuard guard;
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g) th.emplace_back(work, g);
        for (auto &t : th) t.join();
    }
    for (int g = 0; g < G; ++g)
        if (st[(size_t)g] != LDPC_OK) return set_error(st[(size_t)g], "device " + std::to_string(m->dev[(size_t)g]) + " (shard " + std::to_string(g) + "): " + msg[(size_t)g]);
    return LDPC_OK;
}

// ---- ROOT-DEVICE form
ldpc_status ldpc_bp_decode_batch_multi_device(ldpc_bp_multi *m, int64_t batch, const uint8_t *d_syn, uint8_t *d_err,
                                              uint8_t *d_conv, double *d_llr, int32_t *d_iters, void *stream_v)
{
    if (!m) return set_error(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return set_error(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return LDPC_OK;
    if ((m->s > 0 && !d_syn) || (m->n > 0 && !d_err) || !d_conv) return set_error(LDPC_ERR_INVALID_ARGUMENT, "syndromes/errors/converged pointer is NULL");
    const int G = m->ndev;
    const size_t s = (size_t)m->s, n = (size_t)m->n;
    hipStream_t R = (hipStream_t)stream_v;
    DeviceGuard guard;
    const bool use_rccl = m->exchange == LDPC_EXCHANGE_RCCL;
    const bool self = use_rccl && G == 1;                  // the one-GPU rehearsal: the shard travels to itself
    Rccl *N = nullptr;
    if (use_rccl) {
        const ldpc_status cs = ensure_comms(m);
        if (cs != LDPC_OK) return cs;
        N = rccl();
    }
    m->timed = false;
    m->last_batch = batch; m->last_llr = d_llr != nullptr; m->last_iters = d_iters != nullptr;
    // shard buffers of the peers, in their own HBM
    for (int g = self ? 0 : 1; g < G; ++g) {
        int64_t lo, hi;
        shard_bounds(batch, G, g, &lo, &hi);
        const size_t c = (size_t)(hi - lo);
        if (!c) continue;
        MHIP_TRY(hipSetDevice(m->dev[(size_t)g]));
        ldpc_bp_multi::Shard &sh = m->shard[(size_t)g];
        MHIP_TRY(sh.syn.ensure(std::max<size_t>(c * s, 1)));
        MHIP_TRY(sh.err.ensure(std::max<size_t>(c * n, 1)));
        MHIP_TRY(sh.conv.ensure(c));
        if (d_iters) MHIP_TRY(sh.iters.ensure(c * sizeof(int32_t)));
        if (d_llr) MHIP_TRY(sh.llr.ensure(std::max<size_t>(c * n, 1) * sizeof(double)));
    }
    MHIP_TRY(hipSetDevice(m->dev[0]));
    MHIP_TRY(hipEventRecord(m->ev_t[0], R));

    // ---- scatter: syndromes of shard g -> device g
    if (use_rccl) {
        // (nothing but RCCL's own calls can fail inside the group: the device ordinals were checked at create, the shard
        //  buffers are allocated above)
        ncclResult_t nr = ncclSuccess;
        const char *what = "ncclGroupStart (scatter)";
        {
            GroupGuard grp(N);
            nr = grp.start();
            for (int g = self ? 0 : 1; g < G && nr == ncclSuccess; ++g) {
                int64_t lo, hi;
                shard_bounds(batch, G, g, &lo, &hi);
                const size_t c = (size_t)(hi - lo);
                if (!c || !s) continue;
                (void)hipSetDevice(m->dev[0]);
                what = "ncclSend (scatter)";
                nr = N->Send(d_syn + (size_t)lo * s, c * s, ncclUint8, g, m->comm[0], R);
                if (nr != ncclSuccess) break;
                (void)hipSetDevice(m->dev[(size_t)g]);
                what = "ncclRecv (scatter)";
                nr = N->Recv(m->shard[(size_t)g].syn.p, c * s, ncclUint8, 0, m->comm[(size_t)g], self ? R : m->st[(size_t)g]);
            }
            if (grp.open) {
                const ncclResult_t er = grp.end();   // (after a failed call: discards the group and returns that error)
                if (nr == ncclSuccess && er != ncclSuccess) { nr = er; what = "ncclGroupEnd (scatter)"; }
            }
        }
        if (nr != ncclSuccess) {
            (void)set_error(LDPC_ERR_HIP, std::string(what) + ": " + N->GetErrorString(nr));
            exchange_failed(m, R);
            return LDPC_ERR_HIP;
        }
    } else if (G > 1) {
        MHIP_TRY(hipSetDevice(m->dev[0]));
        MHIP_TRY(hipEventRecord(m->ev_in, R));
        for (int g = 1; g < G; ++g) {
            int64_t lo, hi;
            shard_bounds(batch, G, g, &lo, &hi);
            const size_t c = (size_t)(hi - lo);
            if (!c) continue;
            MHIP_TRY(hipSetDevice(m->dev[(size_t)g]));
            MHIP_TRY(hipStreamWaitEvent(m->st[(size_t)g], m->ev_in, 0));
            if (s) MHIP_TRY(hipMemcpyPeerAsync(m->shard[(size_t)g].syn.p, m->dev[(size_t)g], d_syn + (size_t)lo * s, m->dev[0], c * s, m->st[(size_t)g]));
        }
    }
    MHIP_TRY(hipSetDevice(m->dev[0]));
    MHIP_TRY(hipEventRecord(m->ev_t[1], R));

    // ---- decode: every device its shard, on its own stream; the root straight into the caller's arrays
    for (int g = 0; g < G; ++g) {
        int64_t lo, hi;
        shard_bounds(batch, G, g, &lo, &hi);
        const int64_t c = hi - lo;
        if (!c) continue;
        ldpc_status ds;
        if (g == 0 && !self) {
            ds = ldpc_bp_decode_batch_device(m->h[0], c, d_syn + (size_t)lo * s, d_err + (size_t)lo * n, d_conv + lo,
                                             d_llr ? d_llr + (size_t)lo * n : nullptr, d_iters ? d_iters + lo : nullptr, R);
        } else {
            ldpc_bp_multi::Shard &sh = m->shard[(size_t)g];
            ds = ldpc_bp_decode_batch_device(m->h[(size_t)g], c, (const uint8_t *)sh.syn.p, (uint8_t *)sh.err.p, (uint8_t *)sh.conv.p,
                                             d_llr ? (double *)sh.llr.p : nullptr, d_iters ? (int32_t *)sh.iters.p : nullptr,
                                             self ? R : m->st[(size_t)g]);
        }
        if (ds != LDPC_OK) {
            // shards 0 ... g - 1 and the scatter are enqueued, nothing will gather them: wait for them, so that the caller
            // gets its arrays back idle and the shard buffers are free for the next call
            drain_streams(m, R);
            return ds;
        }
    }
    MHIP_TRY(hipSetDevice(m->dev[0]));
    MHIP_TRY(hipEventRecord(m->ev_t[2], R));

    // ---- gather: hard decisions, flags (, iteration counts, LLRs) of shard g -> the caller's arrays on the root
    if (use_rccl) {
        ncclResult_t nr = ncclSuccess;
        const char *what = "ncclGroupStart (gather)";
        {
            GroupGuard grp(N);
            nr = grp.start();
            for (int g = self ? 0 : 1; g < G && nr == ncclSuccess; ++g) {
                int64_t lo, hi;
                shard_bounds(batch, G, g, &lo, &hi);
                const size_t c = (size_t)(hi - lo);
                if (!c) continue;
                ldpc_bp_multi::Shard &sh = m->shard[(size_t)g];
                hipStream_t P = self ? R : m->st[(size_t)g];
                struct Piece { const void *src; void *dst; size_t count; ncclDataType_t ty; };
                const Piece pieces[4] = {
                    {sh.err.p, d_err + (size_t)lo * n, c * n, ncclUint8},
                    {sh.conv.p, d_conv + lo, c, ncclUint8},
                    {d_iters ? sh.iters.p : nullptr, d_iters ? d_iters + lo : nullptr, c, ncclInt32},
                    {d_llr ? sh.llr.p : nullptr, d_llr ? d_llr + (size_t)lo * n : nullptr, c * n, ncclFloat64},
                };
                for (const Piece &q : pieces) {
                    if (!q.src || !q.count) continue;
                    (void)hipSetDevice(m->dev[(size_t)g]);
                    what = "ncclSend (gather)";
                    if ((nr = N->Send(q.src, q.count, q.ty, 0, m->comm[(size_t)g], P)) != ncclSuccess) break;
                    (void)hipSetDevice(m->dev[0]);
                    what = "ncclRecv (gather)";
                    if ((nr = N->Recv(q.dst, q.count, q.ty, g, m->comm[0], R)) != ncclSuccess) break;
                }
            }
            if (grp.open) {
                const ncclResult_t er = grp.end();
                if (nr == ncclSuccess && er != ncclSuccess) { nr = er; what = "ncclGroupEnd (gather)"; }
            }
        }
        if (nr != ncclSuccess) {
            (void)set_error(LDPC_ERR_HIP, std::string(what) + ": " + N->GetErrorString(nr));
            exchange_failed(m, R);
            return LDPC_ERR_HIP;
        }
    } else if (G > 1) {
        for (int g = 1; g < G; ++g) {
            int64_t lo, hi;
            shard_bounds(batch, G, g, &lo, &hi);
            const size_t c = (size_t)(hi - lo);
            if (!c) continue;
            ldpc_bp_multi::Shard &sh = m->shard[(size_t)g];
            MHIP_TRY(hipSetDevice(m->dev[(size_t)g]));
            MHIP_TRY(hipEventRecord(m->ev_done[(size_t)g], m->st[(size_t)g]));
            MHIP_TRY(hipSetDevice(m->dev[0]));
            MHIP_TRY(hipStreamWaitEvent(R, m->ev_done[(size_t)g], 0));
            if (n) MHIP_TRY(hipMemcpyPeerAsync(d_err + (size_t)lo * n, m->dev[0], sh.err.p, m->dev[(size_t)g], c * n, R));
            MHIP_TRY(hipMemcpyPeerAsync(d_conv + lo, m->dev[0], sh.conv.p, m->dev[(size_t)g], c, R));
            if (d_iters) MHIP_TRY(hipMemcpyPeerAsync(d_iters + lo, m->dev[0], sh.iters.p, m->dev[(size_t)g], c * sizeof(int32_t), R));
            if (d_llr && n) MHIP_TRY(hipMemcpyPeerAsync(d_llr + (size_t)lo * n, m->dev[0], sh.llr.p, m->dev[(size_t)g], c * n * sizeof(double), R));
        }
    }
    MHIP_TRY(hipSetDevice(m->dev[0]));
    MHIP_TRY(hipEventRecord(m->ev_t[3], R));
    m->timed = true;
    return LDPC_OK;
}

ldpc_status ldpc_bp_multi_last_status(ldpc_bp_multi *m)
{
    if (!m) return set_error(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    DeviceGuard guard;
    ldpc_status first = LDPC_OK;
    std::string msg;
    for (int g = 0; g < m->ndev; ++g) {
        const ldpc_status st = ldpc_bp_last_status(m->h[(size_t)g]);   // waits for that device's share of the call
        if (st != LDPC_OK && first == LDPC_OK) { first = st; msg = "device " + std::to_string(m->dev[(size_t)g]) + " (shard " + std::to_string(g) + "): " + ldpc_last_error(); }
    }
    if (m->timed) {
        (void)hipSetDevice(m->dev[0]);
        const ldpc_status ws = ldpc_detail::wait_event(m->ev_t[3], m->dev[0], "ldpc_bp_multi_last_status (the gather on the root's stream)");
        if (ws != LDPC_OK && first == LDPC_OK) { first = ws; msg = ldpc_last_error(); }
    }
    return first == LDPC_OK ? LDPC_OK : set_error(first, msg);
}

ldpc_status ldpc_bp_multi_get_info(ldpc_bp_multi *m, ldpc_bp_multi_info *info)
{
    if (!m || !info) return set_error(LDPC_ERR_INVALID_ARGUMENT, "NULL argument");
    std::memset(info, 0, sizeof *info);
    info->ndev = m->ndev;
    info->exchange = m->exchange;
    for (int g = 0; g < m->ndev; ++g) info->devices[g] = m->dev[(size_t)g];
    if (!m->timed) return LDPC_OK;
    DeviceGuard guard;
    MHIP_TRY(hipSetDevice(m->dev[0]));
    {
        const ldpc_status ws = ldpc_detail::wait_event(m->ev_t[3], m->dev[0], "ldpc_bp_multi_get_info (wait for the most recent root-device call)");
        if (ws != LDPC_OK) return ws;
    }
    float a = 0.f, b = 0.f, c = 0.f;
    MHIP_TRY(hipEventElapsedTime(&a, m->ev_t[0], m->ev_t[1]));
    MHIP_TRY(hipEventElapsedTime(&b, m->ev_t[1], m->ev_t[2]));
    MHIP_TRY(hipEventElapsedTime(&c, m->ev_t[2], m->ev_t[3]));
    info->scatter_ms = a; info->root_decode_ms = b; info->gather_ms = c;
    for (int g = 0; g < m->ndev; ++g) {
        int64_t lo, hi;
        shard_bounds(m->last_batch, m->ndev, g, &lo, &hi);
        if (hi <= lo) continue;
        double sweep = 0.0, total = 0.0;
        if (ldpc_bp_last_timing(m->h[(size_t)g], &sweep, &total, nullptr) == LDPC_OK) info->decode_ms_max = std::max(info->decode_ms_max, total);
        if (g > 0) {
            info->scatter_bytes_per_peer = std::max<int64_t>(info->scatter_bytes_per_peer, (hi - lo) * m->s);
            info->gather_bytes_per_peer = std::max<int64_t>(info->gather_bytes_per_peer,
                                                            (hi - lo) * (m->n + 1 + (m->last_iters ? 4 : 0) + (m->last_llr ? 8 * m->n : 0)));
        }
    }
    return LDPC_OK;
}

}  // extern "C"
