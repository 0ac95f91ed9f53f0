// ldpc_bpots.hip -- host side of the BP-OTS decoder: the ldpc_bpots_* entry points
// of include/ldpc_mi355x.h.  Replaces BPOTSDecoder / decode! / batchdecode! of
// src/decoders/bpots_decoder.jl:39-115, 225-340.  Device code: bpots_kernels.hpp.
// Two kernels: LDS-resident (S syndromes per workgroup; every code of the reference's BP-OTS tests) and, for graphs
// whose messages do not fit a CU's LDS, node-parallel with the messages in a global slot (one workgroup per
// syndrome).  Beyond s + 3n + 4s bytes of LDS (n ~ 30,000 for a rate-1/2 code), check degree 32 or bit degree 16:
// LDPC_ERR_UNSUPPORTED.  No CPU path.
#include "../../include/ldpc_mi355x.h"
#include "bpots_kernels.hpp"
#include "host_env.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace ldpc;

#include "host_wait.hpp"   // set_error, and the bounded forms of every host-side wait
using ldpc_detail::set_error;

#define OTS_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError();                                                         \
            return set_error(e_ == hipErrorOutOfMemory ? LDPC_ERR_OUT_OF_MEMORY : LDPC_ERR_HIP, \
                             std::string(#expr) + ": " + hipGetErrorString(e_));             \
        }                                                                                    \
    } while (0)

struct ldpc_bpots_decoder {
    int64_t s = 0, n = 0, nnz = 0, max_iters = 0, T = 9;
    double per = 0.0, C = 2.0;
    int device = 0, num_cus = 0, logS = 0, max_cdeg = 0, max_bdeg = 0;
    int *row_ptr = nullptr, *csc_row = nullptr, *col_ptr = nullptr, *csc2csr = nullptr;
    unsigned int *queue = nullptr;
    void *stage = nullptr;      // device staging for the host-pointer entry
    size_t stage_cap = 0;
    // latency path of the host-pointer entry (DESIGN.md "Latency path"): host-mapped I/O image with a flag word
    unsigned int *done_ctr = nullptr;
    void *lat_pin = nullptr, *lat_pin_dev = nullptr;
    size_t lat_pin_cap = 0;
    unsigned lat_ticket = 0;
    bool kernel_ready = false;  // dynamic-LDS limit set, occupancy known
    int per_cu = 1;
    bool node_mode = false;     // the graph is beyond the LDS kernel: bpots_node_kernel, messages in a global slot
    bool big_mode = false;      // ... and beyond that one too (bytes of a syndrome past the LDS, or nodes wider than the
                                // register buckets): bpots_big_kernel, everything in the global slot, any degree
    double *node_ws = nullptr;  // [grid][slot] of the node kernel
    size_t node_ws_cap = 0;
    ~ldpc_bpots_decoder()
    {
        if (ldpc_detail::device_stalled(device)) return;   // (host_wait.hpp: nothing a stalled device may still use is freed)
        void *all[] = {row_ptr, csc_row, col_ptr, csc2csr, queue, stage, done_ctr, node_ws};
        for (void *q : all)
            if (q) (void)hipFree(q);
        if (lat_pin) (void)hipHostFree(lat_pin);
    }
};

typedef void (*ots_kernel_t)(OtsParams, const int *, const int *, const int *, const int *);

static ots_kernel_t pick_ots(int dc, int dv)
{
    if (dc <= 8) return dv <= 4 ? bpots_lds_kernel<8, 4> : bpots_lds_kernel<8, 16>;
    return dv <= 4 ? bpots_lds_kernel<32, 4> : bpots_lds_kernel<32, 16>;
}

typedef void (*ots_node_kernel_t)(OtsNodeParams, const int *, const int *, const int *, const int *);

static ots_node_kernel_t pick_ots_node(int dc, int dv)
{
    if (dc <= 8) return dv <= 4 ? bpots_node_kernel<8, 4> : bpots_node_kernel<8, 16>;
    return dv <= 4 ? bpots_node_kernel<32, 4> : bpots_node_kernel<32, 16>;
}

extern "C" {

ldpc_status ldpc_bpots_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                              double per, int64_t max_iters, int64_t T, double C, int32_t device,
                              ldpc_bpots_decoder **out)
{
    if (!out) return set_error(LDPC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (s < 0 || n < 0 || nnz < 0 || !colptr || (nnz > 0 && !rowval) || max_iters < 0 || max_iters > INT32_MAX)
        return set_error(LDPC_ERR_INVALID_ARGUMENT, "bad dimensions / NULL pattern / max_iters");
    if (T < 1) return set_error(LDPC_ERR_INVALID_ARGUMENT, "T must be >= 1 (iter % T, bpots_decoder.jl:295)");
    if (colptr[0] != 0 || colptr[n] != nnz) return set_error(LDPC_ERR_INVALID_ARGUMENT, "colptr is not a zero-based CSC pointer array");
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) return set_error(LDPC_ERR_INVALID_ARGUMENT, "colptr is not non-decreasing");
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (rowval[k] < 0 || rowval[k] >= s) return set_error(LDPC_ERR_INVALID_ARGUMENT, "rowval entry outside [0, s)");
            if (k > colptr[j] && rowval[k] <= rowval[k - 1])
                return set_error(LDPC_ERR_INVALID_ARGUMENT, "row indices must be strictly ascending inside each column");
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return set_error(LDPC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    if (device < 0) OTS_TRY(hipGetDevice(&device));
    if (device >= ndev) return set_error(LDPC_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    OTS_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    OTS_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(LDPC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    ldpc_bpots_decoder *d = new (std::nothrow) ldpc_bpots_decoder();
    if (!d) return set_error(LDPC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->s = s; d->n = n; d->nnz = nnz; d->max_iters = max_iters; d->T = T; d->per = per; d->C = C;
    d->device = device; d->num_cus = prop.multiProcessorCount;
    std::vector<int> row_ptr((size_t)s + 1, 0), col_ptr((size_t)n + 1), csc2csr((size_t)std::max<int64_t>(nnz, 1)),
        csc_row((size_t)std::max<int64_t>(nnz, 1));
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
                csc2csr[(size_t)k] = fill[(size_t)rowval[k]]++;
                csc_row[(size_t)k] = (int)rowval[k];
            }
        }
        col_ptr[(size_t)n] = (int)nnz;
    }
    if (nnz >= ((int64_t)1 << 28) || s >= ((int64_t)1 << 28) || n >= ((int64_t)1 << 28)) {
        delete d;
        return set_error(LDPC_ERR_UNSUPPORTED, "BP-OTS kernels: graph too large for 32-bit edge indexing");
    }
    d->logS = -1;
    // tests (experiments build, read at create): LDPC_BPOTS_FORCE_NODE = 1 sends small graphs through the node kernel, 2 through the unlimited one
    const int force_node = exp_env("LDPC_BPOTS_FORCE_NODE") ? std::max(1, std::atoi(exp_env("LDPC_BPOTS_FORCE_NODE"))) : 0;
    const bool wide = d->max_cdeg > 32 || d->max_bdeg > 16;   // beyond the register buckets of the two fast kernels
    if (!force_node && !wide && nnz < 65535 && s < 65535 && n < 65535)   // (uint16 graph copies in LDS)
        for (int l = 0; l <= 6; ++l) {
            const size_t b = ots_lds_bytes((int)s, (int)n, (int)nnz, 1 << l) + 8192;
            if (b <= (size_t)76 * 1024 || (d->logS < 0 && b <= (size_t)156 * 1024)) d->logS = l;
            else if (b > (size_t)156 * 1024) break;
        }
    if (d->logS < 0) {
        // beyond the LDS kernel: one workgroup per syndrome, messages in a global slot, bytes / parities in LDS
        d->node_mode = true;
        // ... and when even the bytes and parities of one syndrome do not fit the LDS, or nodes are wider than the register
        // buckets: everything in the global slot, any degree (the reference's BPOTSDecoder has no limit either)
        d->big_mode = wide || force_node >= 2 || ots_node_lds_bytes((int)s, (int)n) + 1024 > (size_t)156 * 1024;
    }
    auto up = [&](int *&dst, const std::vector<int> &v) -> bool {
        if (hipMalloc((void **)&dst, std::max<size_t>(v.size(), 1) * sizeof(int)) != hipSuccess) return false;
        return hipMemcpy(dst, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(d->row_ptr, row_ptr) || !up(d->csc_row, csc_row) || !up(d->col_ptr, col_ptr) || !up(d->csc2csr, csc2csr) ||
        hipMalloc((void **)&d->queue, 64) != hipSuccess || hipMalloc((void **)&d->done_ctr, 64) != hipSuccess ||
        hipMemset(d->done_ctr, 0, 64) != hipSuccess) {
        (void)hipGetLastError();
        delete d;
        return set_error(LDPC_ERR_OUT_OF_MEMORY, "device allocation of the Tanner graph failed");
    }
    *out = d;
    return LDPC_OK;
}

int32_t ldpc_bpots_kernel(const ldpc_bpots_decoder *d) { return d ? (d->big_mode ? 5 : d->node_mode ? 3 : 2) : 0; }

ldpc_status ldpc_bpots_destroy(ldpc_bpots_decoder *d)
{
    if (!d) return LDPC_OK;
    (void)hipSetDevice(d->device);
    const ldpc_status st = ldpc_detail::wait_device(d->device, "ldpc_bpots_destroy (device synchronise)");
    delete d;
    return st;
}

}  // extern "C"

struct OtsLatencyCtl {
    unsigned int *flag;
    unsigned int ticket;
};

static ldpc_status bpots_decode_impl(ldpc_bpots_decoder *d, int64_t batch, const uint8_t *d_syn, uint8_t *d_err,
                                     uint8_t *d_conv, int32_t *d_iters, void *stream_v, const OtsLatencyCtl *lat)
{
    if (!d) return set_error(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return set_error(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return LDPC_OK;
    if ((d->s > 0 && !d_syn) || (d->n > 0 && !d_err) || !d_conv) return set_error(LDPC_ERR_INVALID_ARGUMENT, "NULL batch pointer");
    hipStream_t stream = (hipStream_t)stream_v;
    OTS_TRY(hipSetDevice(d->device));
    if (ldpc_detail::device_stalled(d->device)) return ldpc_detail::stalled_error(d->device);
    if (d->max_iters == 0) {   // the loop at :239 never runs: best_decisions = 0, converged = false
        if (d->n > 0) OTS_TRY(hipMemsetAsync(d_err, 0, (size_t)batch * d->n, stream));
        OTS_TRY(hipMemsetAsync(d_conv, 0, (size_t)batch, stream));
        if (d_iters) OTS_TRY(hipMemsetAsync(d_iters, 0, (size_t)batch * sizeof(int32_t), stream));
        return LDPC_OK;
    }
    if (d->node_mode) {
        if (batch > (1ll << 30)) return set_error(LDPC_ERR_UNSUPPORTED, "batch too large for one call");
        ots_node_kernel_t nk = d->big_mode ? (ots_node_kernel_t)bpots_big_kernel : pick_ots_node(d->max_cdeg, d->max_bdeg);
        const size_t nlds = d->big_mode ? 0 : ots_node_lds_bytes((int)d->s, (int)d->n);
        if (!d->kernel_ready) {
            if (nlds) OTS_TRY(hipFuncSetAttribute((const void *)nk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nlds));
            d->kernel_ready = true;
        }
        const int grid = (int)std::min<int64_t>(batch, d->num_cus);
        const size_t slot = d->big_mode ? ots_big_slot_doubles((int)d->s, (int)d->n, (int)d->nnz) : ots_node_slot_doubles((int)d->n, (int)d->nnz);
        if (d->node_ws_cap < (size_t)grid * slot * sizeof(double)) {
            if (d->node_ws) {
                const ldpc_status ws = ldpc_detail::wait_device(d->device, "BP-OTS workspace regrow (device synchronise before the free)");
                if (ws != LDPC_OK) return ws;
                (void)hipFree(d->node_ws);
            }
            d->node_ws = nullptr; d->node_ws_cap = 0;
            OTS_TRY(hipMalloc((void **)&d->node_ws, (size_t)grid * slot * sizeof(double)));
            d->node_ws_cap = (size_t)grid * slot * sizeof(double);
        }
        OtsNodeParams np{};
        np.s = (int)d->s; np.n = (int)d->n; np.nnz = (int)d->nnz; np.max_iters = (int)d->max_iters; np.T = (int)d->T;
        np.batch = batch;
        np.prior = std::log((1 - (2 * d->per / 3)) / (2 * d->per / 3));   // bpots_decoder.jl:231
        np.C = d->C;
        np.syn = d_syn; np.err = d_err; np.conv = d_conv; np.iters = d_iters; np.queue = d->queue;
        np.ws = d->node_ws; np.slot_doubles = (long long)slot;
        OTS_TRY(hipMemsetAsync(d->queue, 0, 64, stream));
        hipLaunchKernelGGL(nk, dim3((unsigned)grid), dim3(kOtsNodeThreads), nlds, stream, np, (const int *)d->row_ptr,
                           (const int *)d->csc_row, (const int *)d->col_ptr, (const int *)d->csc2csr);
        OTS_TRY(hipGetLastError());
        return LDPC_OK;
    }
    const int64_t ngroups = (batch + (1ll << d->logS) - 1) >> d->logS;
    if (ngroups > (1ll << 30)) return set_error(LDPC_ERR_UNSUPPORTED, "batch too large for one call");
    OtsParams p;
    p.s = (int)d->s; p.n = (int)d->n; p.nnz = (int)d->nnz; p.max_iters = (int)d->max_iters; p.T = (int)d->T;
    p.logS = d->logS; p.ngroups = (int)ngroups; p.batch = batch;
    p.prior = std::log((1 - (2 * d->per / 3)) / (2 * d->per / 3));   // bpots_decoder.jl:231
    p.C = d->C;
    p.syn = d_syn; p.err = d_err; p.conv = d_conv; p.iters = d_iters; p.queue = d->queue;
    const size_t lds = ots_lds_bytes((int)d->s, (int)d->n, (int)d->nnz, 1 << d->logS);
    ots_kernel_t k = pick_ots(d->max_cdeg, d->max_bdeg);
    if (!d->kernel_ready) {
        OTS_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, kOtsThreads, lds) != hipSuccess || per_cu <= 0) {
            (void)hipGetLastError();
            per_cu = 1;
        }
        d->per_cu = per_cu;
        d->kernel_ready = true;
    }
    p.done_count = d->done_ctr; p.done_flag = nullptr; p.done_ticket = 0;
    if (lat) {   // one workgroup per group, no queue to reset: the launch is the only runtime call
        p.queue = nullptr; p.chunk = 1;
        p.done_flag = lat->flag; p.done_ticket = lat->ticket;
        hipLaunchKernelGGL(k, dim3((unsigned)ngroups), dim3(kOtsThreads), lds, stream, p, (const int *)d->row_ptr,
                           (const int *)d->csc_row, (const int *)d->col_ptr, (const int *)d->csc2csr);
        OTS_TRY(hipGetLastError());
        return LDPC_OK;
    }
    const int grid = (int)std::min<int64_t>(ngroups, (int64_t)d->per_cu * d->num_cus);
    p.chunk = (int)std::max<int64_t>(1, std::min<int64_t>(64, ngroups / ((int64_t)grid * 16)));
    OTS_TRY(hipMemsetAsync(d->queue, 0, 64, stream));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kOtsThreads), lds, stream, p, (const int *)d->row_ptr,
                       (const int *)d->csc_row, (const int *)d->col_ptr, (const int *)d->csc2csr);
    OTS_TRY(hipGetLastError());
    return LDPC_OK;
}

extern "C" {

ldpc_status ldpc_bpots_decode_batch_device(ldpc_bpots_decoder *d, int64_t batch, const uint8_t *d_syn, uint8_t *d_err,
                                           uint8_t *d_conv, int32_t *d_iters, void *stream_v)
{
    return bpots_decode_impl(d, batch, d_syn, d_err, d_conv, d_iters, stream_v, nullptr);
}

ldpc_status ldpc_bpots_decode_batch(ldpc_bpots_decoder *d, int64_t batch, const uint8_t *syn, uint8_t *err,
                                    uint8_t *conv, int32_t *iters)
{
    if (!d) return set_error(LDPC_ERR_INVALID_ARGUMENT, "decoder is NULL");
    if (batch < 0) return set_error(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return LDPC_OK;
    if ((d->s > 0 && !syn) || (d->n > 0 && !err) || !conv) return set_error(LDPC_ERR_INVALID_ARGUMENT, "NULL batch pointer");
    OTS_TRY(hipSetDevice(d->device));
    const size_t s = (size_t)d->s, n = (size_t)d->n, B = (size_t)batch;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_err = up(B * s), o_conv = o_err + up(B * n), o_it = o_conv + up(B), total = o_it + up(B * 4);
    static const bool lat_off = exp_env("LDPC_NO_LATENCY_PATH") != nullptr;
    const int64_t lat_groups = d->node_mode ? batch : (batch + (1ll << d->logS) - 1) >> d->logS;
    if (!d->node_mode && !lat_off && total <= ((size_t)256 << 10) && d->max_iters > 0 && lat_groups <= 2 * (int64_t)d->num_cus) {
        const size_t hdr = 256;
        if (!d->lat_pin) {
            const size_t cap = hdr + ((size_t)256 << 10);
            OTS_TRY(hipHostMalloc(&d->lat_pin, cap, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(d->lat_pin, 0, hdr);
            OTS_TRY(hipHostGetDevicePointer(&d->lat_pin_dev, d->lat_pin, 0));
            d->lat_pin_cap = cap;
        }
        char *hp = (char *)d->lat_pin + hdr, *dp = (char *)d->lat_pin_dev + hdr;
        volatile unsigned int *flag = (volatile unsigned int *)d->lat_pin;
        if (++d->lat_ticket == 0) d->lat_ticket = 1;
        const OtsLatencyCtl lc{(unsigned int *)d->lat_pin_dev, d->lat_ticket};
        std::memcpy(hp, syn, B * s);
        ldpc_status lst = bpots_decode_impl(d, batch, (const uint8_t *)dp, (uint8_t *)(dp + o_err), (uint8_t *)(dp + o_conv),
                                            (int32_t *)(dp + o_it), nullptr, &lc);
        if (lst != LDPC_OK) return lst;
        const auto lat_t0 = std::chrono::steady_clock::now();
        for (uint64_t spins = 1;; ++spins) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == lc.ticket) break;
            if ((spins & 0xffff) == 0) {   // every ~65k polls: is the kernel still alive?  (and the bound of host_wait.hpp)
                const int64_t lim = ldpc_detail::wait_limit_ms();
                if (lim > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - lat_t0).count() > lim)
                    return ldpc_detail::wait_stream(nullptr, d->device, "BP-OTS latency path (flag of the last workgroup)");   // (expires at once: names the wait, marks the device)
                const hipError_t q = hipStreamQuery(nullptr);
                if (q == hipSuccess) {
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == lc.ticket) break;
                    return set_error(LDPC_ERR_HIP, "latency path: the kernel finished without raising its flag");
                }
                if (q != hipErrorNotReady) {
                    (void)hipGetLastError();
                    return set_error(LDPC_ERR_HIP, std::string("latency path: ") + hipGetErrorString(q));
                }
            }
            __builtin_ia32_pause();
        }
        std::memcpy(err, hp + o_err, B * n);
        std::memcpy(conv, hp + o_conv, B);
        if (iters) std::memcpy(iters, hp + o_it, B * sizeof(int32_t));
        return LDPC_OK;
    }
    if (d->stage_cap < total) {
        if (d->stage) {
            const ldpc_status ws = ldpc_detail::wait_device(d->device, "BP-OTS staging regrow (device synchronise before the free)");
            if (ws != LDPC_OK) return ws;
            (void)hipFree(d->stage);
        }
        d->stage = nullptr; d->stage_cap = 0;
        OTS_TRY(hipMalloc(&d->stage, total));
        d->stage_cap = total;
    }
    char *dp = (char *)d->stage;
    if (s > 0) OTS_TRY(hipMemcpyAsync(dp, syn, B * s, hipMemcpyHostToDevice, nullptr));
    ldpc_status st = ldpc_bpots_decode_batch_device(d, batch, (const uint8_t *)dp, (uint8_t *)(dp + o_err),
                                                    (uint8_t *)(dp + o_conv), (int32_t *)(dp + o_it), nullptr);
    if (st != LDPC_OK) return st;
    if (n > 0) OTS_TRY(hipMemcpyAsync(err, dp + o_err, B * n, hipMemcpyDeviceToHost, nullptr));
    OTS_TRY(hipMemcpyAsync(conv, dp + o_conv, B, hipMemcpyDeviceToHost, nullptr));
    if (iters) OTS_TRY(hipMemcpyAsync(iters, dp + o_it, B * 4, hipMemcpyDeviceToHost, nullptr));
    return ldpc_detail::wait_stream(nullptr, d->device, "ldpc_bpots_decode_batch (stream synchronise)");
}

}  // extern "C"
