// vmm_probe3.hip -- third placement experiment.  vmm_probe2 showed (before it faulted while re-mapping a VA range
// it had used before at another offset) that with the physical memory held FIXED the in-place streaming pattern
// runs 5.4 ... 6.6 TB/s depending on the virtual base alone, in steps of 32 MiB.  Here, carefully:
//   * ONE physical allocation (hipMemCreate of the whole workspace), mapped by ONE hipMemMap and unmapped by one
//     matching hipMemUnmap; every candidate base lives in a FRESH reservation that is never mapped twice
//     (all reservations are held to the end);
//   * the patterns of the real kernel: sweeps start at a per-workgroup rotation (LDPC_ROTATE), check sweep =
//     8 contiguous rows in place, variable sweep = 4 scattered rows in place;
//   * map / unmap are timed (what a placement search inside the library would cost);
//   * plain hipMalloc candidates at the end for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe3 tools/vmm_probe3.hip
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
    const int ncand = argc > 1 ? atoi(argv[1]) : 40;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    const size_t pad = 1053184;
    const size_t stride = ((size_t)rows * 512 + pad) / 8;
    const size_t ws = ((size_t)slots * stride * 8 + 2 * MiB - 1) / (2 * MiB) * (2 * MiB);
    double t0 = now_ms();
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, ws, &prop, 0));
    printf("hipMemCreate of %.2f GiB: %.1f ms\n", ws / 1073741824.0, now_ms() - t0);
    std::vector<void *> resv;
    bool zeroed = false;
    for (int c = 0; c < ncand; ++c) {
        // candidate c: fresh 1 GiB-aligned reservation, base at offset (c mod 32) x 32 MiB (+ (c / 32) x 1 GiB)
        const size_t off = (size_t)(c % 32) * 32 * MiB + (size_t)(c / 32) * GiB;
        void *rv = nullptr;
        CK(hipMemAddressReserve(&rv, ws + 4 * GiB, GiB, nullptr, 0));
        resv.push_back(rv);
        char *base = (char *)(((uintptr_t)rv + GiB - 1) & ~(uintptr_t)(GiB - 1)) + off;
        t0 = now_ms();
        CK(hipMemMap(base, ws, 0, h, 0));
        CK(hipMemSetAccess(base, ws, &acc, 1));
        const double t_map = now_ms() - t0;
        if (!zeroed) { CK(hipMemset(base, 0, ws)); CK(hipDeviceSynchronize()); zeroed = true; }
        double v = 0, cr = 0, cl = 0;
        for (int rep = 0; rep < 2; ++rep) {
            v = std::max(v, run(0, (double *)base, stride, 3));
            cr = std::max(cr, run(1, (double *)base, stride, 3));
            cl = std::max(cl, run(2, (double *)base, stride, 3));
        }
        t0 = now_ms();
        CK(hipMemUnmap(base, ws));
        const double t_unmap = now_ms() - t0;
        printf("cand %2d base %p (resv %p, 1GiB-aligned + %4zu MiB): var %.2f  check(rot) %.2f  check(lockstep) %.2f TB/s   map %.1f ms unmap %.1f ms\n",
               c, (void *)base, rv, off / MiB, v, cr, cl, t_map, t_unmap);
        fflush(stdout);
    }
    CK(hipMemRelease(h));
    for (int c = 0; c < 4; ++c) {
        void *q = nullptr;
        if (hipMalloc(&q, ws) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(q, 0, ws));
        double v = 0, cr = 0, cl = 0;
        for (int rep = 0; rep < 2; ++rep) {
            v = std::max(v, run(0, (double *)q, stride, 3));
            cr = std::max(cr, run(1, (double *)q, stride, 3));
            cl = std::max(cl, run(2, (double *)q, stride, 3));
        }
        printf("hipMalloc candidate %d @%p: var %.2f  check(rot) %.2f  check(lockstep) %.2f TB/s\n", c, q, v, cr, cl);
    }
    for (void *rv : resv) (void)rv;   // reservations are held until the process ends
    return 0;
}
