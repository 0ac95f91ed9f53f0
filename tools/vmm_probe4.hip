// vmm_probe4.hip -- ONE workspace configuration per PROCESS (vmm_probe2/3 re-mapped memory inside one process; the
// runtime does not survive that reliably, and the numbers may have come from stale translations).  A clean answer
// to "what makes a C3-size workspace fast": how the PHYSICAL memory was obtained (hipMalloc, one hipMemCreate, chunks
// of 2 MiB ... 1 GiB, in creation order or shuffled) and where it is mapped (offset of the virtual base).
// Usage: tools/vmm_probe4 <malloc | chunk_MiB (0 = one handle)> [shuffle 0/1] [va_offset_MiB] [hold_GiB]
//   hold_GiB: a hipMalloc of that size made (and kept) first, so that the physical memory comes from elsewhere
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe4 tools/vmm_probe4.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) var_like(double *base, size_t slot_stride, int rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    const unsigned r = (unsigned)rows, nb = r / 4u;
    const unsigned rot = (blockIdx.x * 2246822519u) % nb;
    for (int it = 0; it < iters; ++it) {
        for (unsigned j0 = (unsigned)w; j0 < nb; j0 += 8u) {
            const unsigned j = (j0 + rot >= nb) ? j0 + rot - nb : j0 + rot;
            const unsigned a = (j * 2654435761u + 12345u) % r, b = (j * 2246822519u + 977u) % r,
                           c = (j * 3266489917u + 31u) % r, d = (j * 668265263u + 7u) % r;
            const double v0 = M[(size_t)a * 64], v1 = M[(size_t)b * 64], v2 = M[(size_t)c * 64], v3 = M[(size_t)d * 64];
            M[(size_t)a * 64] = v1; M[(size_t)b * 64] = v2; M[(size_t)c * 64] = v3; M[(size_t)d * 64] = v0;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) check_like(double *base, size_t slot_stride, int rows, int iters, int rotate)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    const int nc = rows / 8;
    const int rot = rotate ? (int)((blockIdx.x * 2654435761u) % (unsigned)nc) : 0;
    for (int it = 0; it < iters; ++it) {
        for (int i0 = w; i0 < nc; i0 += 8) {
            const int i = (i0 + rot >= nc) ? i0 + rot - nc : i0 + rot;
            double *R = M + (size_t)i * 8 * 64;
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = R[k * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k) R[k * 64] = v[k] * 1.0000001;
        }
        __syncthreads();
    }
}

static hipEvent_t ea, eb;
static const int slots = 768, rows = 65536;

static double run(int kind, double *base, size_t stride_doubles, int iters)   // 0 var-like, 1 check-like rotated, 2 check-like lockstep
{
    auto launch = [&](int n) {
        if (kind == 0) hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, n);
        else hipLaunchKernelGGL(check_like, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, n, kind == 1 ? 1 : 0);
    };
    launch(1);
    CK(hipEventRecord(ea));
    launch(iters);
    CK(hipEventRecord(eb));
    CK(hipEventSynchronize(eb));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, ea, eb));
    return 2.0 * (double)slots * rows * 512.0 * iters / (ms * 1e-3) / 1e12;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "malloc";
    const int shuffle = argc > 2 ? atoi(argv[2]) : 0;
    const size_t va_off_mib = argc > 3 ? (size_t)atoll(argv[3]) : 0;
    const size_t hold_gib = argc > 4 ? (size_t)atoll(argv[4]) : 0;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    const size_t pad = 1053184;
    const size_t stride = ((size_t)rows * 512 + pad) / 8;
    void *hold = nullptr;
    if (hold_gib) CK(hipMalloc(&hold, hold_gib * GiB));
    char *base = nullptr;
    double t0 = now_ms();
    if (!strcmp(mode, "malloc")) {
        const size_t ws = (size_t)slots * stride * 8;
        CK(hipMalloc((void **)&base, ws));
        CK(hipMemset(base, 0, ws));
    } else {
        size_t chunk = (size_t)atoll(mode) * MiB;
        const size_t unit = chunk ? chunk : 2 * MiB;
        const size_t ws = ((size_t)slots * stride * 8 + unit - 1) / unit * unit;
        if (!chunk) chunk = ws;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        const size_t nchunk = ws / chunk;
        std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
        for (size_t k = 0; k < nchunk; ++k) CK(hipMemCreate(&h[k], chunk, &prop, 0));
        std::vector<size_t> order(nchunk);
        for (size_t k = 0; k < nchunk; ++k) order[k] = k;
        if (shuffle) {
            unsigned long long sd = 88172645463325252ull;
            for (size_t i = nchunk - 1; i > 0; --i) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(order[i], order[sd % (i + 1)]); }
        }
        void *rv = nullptr;
        CK(hipMemAddressReserve(&rv, ws + 4 * GiB, 0, nullptr, 0));
        base = (char *)(((uintptr_t)rv + GiB - 1) & ~(uintptr_t)(GiB - 1)) + va_off_mib * MiB;
        for (size_t k = 0; k < nchunk; ++k) CK(hipMemMap(base + k * chunk, chunk, 0, h[order[k]], 0));
        CK(hipMemSetAccess(base, ws, &acc, 1));
        CK(hipMemset(base, 0, ws));
    }
    CK(hipDeviceSynchronize());
    const double t_setup = now_ms() - t0;
    double v = 0, cr = 0, cl = 0;
    for (int rep = 0; rep < 3; ++rep) {
        v = std::max(v, run(0, (double *)base, stride, 3));
        cr = std::max(cr, run(1, (double *)base, stride, 3));
        cl = std::max(cl, run(2, (double *)base, stride, 3));
    }
    printf("%-8s shuffle %d va+%4zu MiB hold %3zu GiB @%p: var %.2f  check(rot) %.2f  check(lockstep) %.2f TB/s   (setup %.0f ms)\n",
           mode, shuffle, va_off_mib, hold_gib, (void *)base, v, cr, cl, t_setup);
    return 0;   // no unmap, no free: the process ends
}
