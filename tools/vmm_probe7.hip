// vmm_probe7.hip -- can 1 GiB physical chunks be GRADED and the good ones kept?  (vmm_probe6: under full load a slow
// workspace is slow region by region, and regions differ.)  50 chunks; set A = chunks 0-24 and set B = 25-49, each
// mapped once at a 1 GiB-aligned base of its own and graded per chunk (mean ticks of the slots that lie in it,
// slots dealt so that every chunk holds early and late workgroups alike); then set C = the 25 best chunks, mapped at
// a fresh base (after A and B are unmapped): is it at least as fast as the better of A and B, and do the chunks
// keep their grades?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe7 tools/vmm_probe7.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) var_like(double *base, size_t slot_stride, int rows, int iters, unsigned long long *ticks)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned slot = (blockIdx.x * 37u) % gridDim.x;
    double *M = base + (size_t)slot * slot_stride + lane;
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
    if (threadIdx.x == 0 && ticks) ticks[slot] = wall_clock64() - t0;
}

static const int slots = 768, rows = 65536;
static const size_t GiB = (size_t)1 << 30;

int main()
{
    CK(hipSetDevice(0));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const size_t pad = 1053184;
    const size_t stride = ((size_t)rows * 512 + pad) / 8;
    const size_t K = ((size_t)slots * stride * 8 + GiB - 1) / GiB;   // chunks per workspace (25)
    unsigned long long *d_ticks;
    CK(hipMalloc((void **)&d_ticks, slots * 8));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t N = 2 * K;
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    for (size_t k = 0; k < N; ++k) CK(hipMemCreate(&h[k], GiB, &prop, 0));
    std::vector<double> grade(N, 0.0);
    auto run_set = [&](const char *name, const std::vector<size_t> &set, bool record) {
        void *rv; CK(hipMemAddressReserve(&rv, K * GiB + GiB, 0, nullptr, 0));
        char *base = (char *)(((uintptr_t)rv + GiB - 1) & ~(uintptr_t)(GiB - 1));
        for (size_t q = 0; q < K; ++q) CK(hipMemMap(base + q * GiB, GiB, 0, h[set[q]], 0));
        CK(hipMemSetAccess(base, K * GiB, &acc, 1));
        hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, (double *)base, stride, rows, 1, (unsigned long long *)nullptr);
        CK(hipEventRecord(ea));
        hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, (double *)base, stride, rows, 3, d_ticks);
        CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb));
        float ms; CK(hipEventElapsedTime(&ms, ea, eb));
        std::vector<unsigned long long> t(slots);
        CK(hipMemcpy(t.data(), d_ticks, slots * 8, hipMemcpyDeviceToHost));
        printf("%-12s @%p: %.2f TB/s; mean kiloticks per chunk:", name, (void *)base, 2.0 * slots * rows * 512.0 * 3 / (ms * 1e-3) / 1e12);
        for (size_t q = 0; q < K; ++q) {
            double sum = 0; int cnt = 0;
            for (int k = 0; k < slots; ++k) if (((size_t)k * stride * 8 + stride * 4) / GiB == q) { sum += (double)t[k]; ++cnt; }
            const double g = cnt ? sum / cnt : 0.0;
            if (record) grade[set[q]] = g;
            printf(" %zu:%.0f", set[q], g / 1e3);
        }
        printf("\n"); fflush(stdout);
        for (size_t q = 0; q < K; ++q) CK(hipMemUnmap(base + q * GiB, GiB));
    };
    std::vector<size_t> A(K), B(K);
    std::iota(A.begin(), A.end(), 0); std::iota(B.begin(), B.end(), K);
    // first touch
    run_set("A (touch)", A, false); run_set("B (touch)", B, false);
    run_set("A", A, true); run_set("B", B, true);
    run_set("A again", A, false);
    std::vector<size_t> order(N);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return grade[a] < grade[b]; });
    std::vector<size_t> C(order.begin(), order.begin() + K), D(order.end() - K, order.end());
    run_set("C = best 25", C, false);
    run_set("D = worst 25", D, false);
    std::reverse(C.begin(), C.end());
    run_set("C reversed", C, false);
    return 0;
}
