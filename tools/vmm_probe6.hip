// vmm_probe6.hip -- WHERE inside a slow workspace is the time lost?  Every workgroup records the 100 MHz ticks its
// own sweep took; printed per slot class: if a slow allocation is slow in all slots alike the cause is global
// (translation reach, address hashing); if a few 1 GiB regions are slow, chunks can be graded and swapped.
// One process: hipMalloc candidates A, B (held), then 1 GiB-chunk workspaces C, D (held).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe6 tools/vmm_probe6.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) var_like(double *base, size_t slot_stride, int rows, int iters, unsigned long long *ticks)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    const unsigned r = (unsigned)rows, nb = r / 4u;
    const unsigned rot = (blockIdx.x * 2246822519u) % nb;
    const unsigned long long t0 = wall_clock64();
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
    if (threadIdx.x == 0 && ticks) ticks[blockIdx.x] = wall_clock64() - t0;
}

static const int slots = 768, rows = 65536;

int main()
{
    CK(hipSetDevice(0));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t GiB = (size_t)1 << 30;
    const size_t pad = 1053184;
    const size_t stride = ((size_t)rows * 512 + pad) / 8;
    const size_t ws = ((size_t)slots * stride * 8 + GiB - 1) / GiB * GiB;
    unsigned long long *d_ticks;
    CK(hipMalloc((void **)&d_ticks, slots * 8));
    std::vector<unsigned long long> ticks(slots);
    auto grade = [&](const char *name, double *base) {
        hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, base, stride, rows, 1, (unsigned long long *)nullptr);
        CK(hipEventRecord(ea));
        hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, base, stride, rows, 3, d_ticks);
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        CK(hipMemcpy(ticks.data(), d_ticks, slots * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> s = ticks; std::sort(s.begin(), s.end());
        printf("%-22s @%p: %.2f TB/s; per-slot ticks min %llu p10 %llu median %llu p90 %llu max %llu\n", name, (void *)base,
               2.0 * slots * rows * 512.0 * 3 / (ms * 1e-3) / 1e12, s[0], s[slots / 10], s[slots / 2], s[slots * 9 / 10], s[slots - 1]);
        // by GiB of the workspace
        printf("   mean ticks per GiB region:");
        const int nreg = (int)(ws / GiB);
        for (int g = 0; g < nreg; ++g) {
            double sum = 0; int cnt = 0;
            for (int k = 0; k < slots; ++k) if ((int)(((size_t)k * stride * 8 + stride * 4) / GiB) == g) { sum += (double)ticks[k]; ++cnt; }
            printf(" %.0f", cnt ? sum / cnt : 0.0);
        }
        printf("\n");
        fflush(stdout);
    };
    for (int c = 0; c < 3; ++c) {
        double *q; CK(hipMalloc((void **)&q, ws)); CK(hipMemset(q, 0, ws));
        char nm[32]; snprintf(nm, sizeof nm, "hipMalloc %d", c);
        grade(nm, q); grade(nm, q);
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int c = 0; c < 3; ++c) {
        void *rv; CK(hipMemAddressReserve(&rv, ws + GiB, 0, nullptr, 0));
        char *base = (char *)(((uintptr_t)rv + GiB - 1) & ~(uintptr_t)(GiB - 1));
        for (size_t k = 0; k < ws / GiB; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, GiB, &prop, 0)); CK(hipMemMap(base + k * GiB, GiB, 0, h, 0)); }
        CK(hipMemSetAccess(base, ws, &acc, 1));
        CK(hipMemset(base, 0, ws));
        char nm[32]; snprintf(nm, sizeof nm, "1 GiB chunks %d", c);
        grade(nm, (double *)base); grade(nm, (double *)base);
    }
    return 0;
}
