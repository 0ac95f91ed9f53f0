// vmm_probe.hip -- WHY does the speed of the message sweeps depend on where the workspace lands in HBM?
// (DESIGN.md "Workspace placement": the same kernels run 4.9 ... 5.6 TB/s on different 24.8 GiB
// allocations, reproducibly per allocation.)  hipMalloc hands out virtual AND physical placement at
// once; the virtual-memory API separates them:
//   1. create N physical chunks (hipMemCreate), map them side by side, probe each chunk alone with the
//      whole chip (768 workgroups, variable-sweep pattern = random 512-byte row gathers/scatters, and
//      check-sweep pattern = in-place streaming)  -> is "fast" a property of a physical chunk?
//   2. build 25-chunk workspaces (768 slots x 32 MiB + pad, the C3 geometry) out of chosen chunks --
//      consecutive windows (what hipMalloc candidates are), the fastest / slowest chunks of step 1,
//      the same chunks at another virtual address, the same chunks in shuffled order -- and run the
//      real slot geometry on each  -> physical set, virtual address or order?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe tools/vmm_probe.hip
// Run:   tools/vmm_probe [total_GiB=150] [chunk_MiB=1024]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) rnd_sweep(double *base, size_t slot_stride, int rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    const unsigned r = (unsigned)rows;
    for (int it = 0; it < iters; ++it) {
        for (unsigned j = (unsigned)w; j < r / 4u; j += 8u) {
            const unsigned a = (j * 2654435761u + 12345u) % r, b = (j * 2246822519u + 977u) % r,
                           c = (j * 3266489917u + 31u) % r, d = (j * 668265263u + 7u) % r;
            const double v0 = M[(size_t)a * 64], v1 = M[(size_t)b * 64], v2 = M[(size_t)c * 64], v3 = M[(size_t)d * 64];
            M[(size_t)a * 64] = v1; M[(size_t)b * 64] = v2; M[(size_t)c * 64] = v3; M[(size_t)d * 64] = v0;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(512) seq_sweep(double *base, size_t slot_stride, int rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    for (int it = 0; it < iters; ++it) {
        for (int i = w; i < rows / 8; i += 8) {
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

// TB/s (read + write) of `iters` passes of one pattern over `slots` slots of `rows` rows
static double run(bool rnd, double *base, size_t stride_doubles, int slots, int rows, int iters)
{
    if (rnd) hipLaunchKernelGGL(rnd_sweep, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, 1);
    else hipLaunchKernelGGL(seq_sweep, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, 1);
    CK(hipEventRecord(ea));
    if (rnd) hipLaunchKernelGGL(rnd_sweep, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, iters);
    else hipLaunchKernelGGL(seq_sweep, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, iters);
    CK(hipEventRecord(eb));
    CK(hipEventSynchronize(eb));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, ea, eb));
    return 2.0 * (double)slots * rows * 512.0 * iters / (ms * 1e-3) / 1e12;
}

int main(int argc, char **argv)
{
    const size_t total_gib = argc > 1 ? (size_t)atoll(argv[1]) : 150;
    const size_t chunk_mib = argc > 2 ? (size_t)atoll(argv[2]) : 1024;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gmin = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    size_t chunk = chunk_mib << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    size_t free_b = 0, tot_b = 0;
    CK(hipMemGetInfo(&free_b, &tot_b));
    printf("granularity recommended %zu min %zu; chunk %zu MiB; HBM free %.1f GiB of %.1f\n", gran, gmin, chunk >> 20,
           free_b / 1073741824.0, tot_b / 1073741824.0);
    const size_t want = std::min((total_gib << 30) / chunk, (free_b - ((size_t)40 << 30)) / chunk);
    std::vector<hipMemGenericAllocationHandle_t> h;
    for (size_t k = 0; k < want; ++k) {
        hipMemGenericAllocationHandle_t q;
        if (hipMemCreate(&q, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        h.push_back(q);
    }
    const size_t N = h.size();
    printf("%zu physical chunks created\n", N);
    fflush(stdout);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, N * chunk, 0, nullptr, 0));
    for (size_t k = 0; k < N; ++k) CK(hipMemMap((char *)va + k * chunk, chunk, 0, h[k], 0));
    CK(hipMemSetAccess(va, N * chunk, &acc, 1));
    CK(hipMemset(va, 0, N * chunk));
    CK(hipDeviceSynchronize());
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));

    // ---- step 1: every chunk alone
    const int slots = 768;
    const size_t c_stride = chunk / slots / 512 * 64;   // doubles per sub-slot (whole rows)
    const int c_rows = (int)(c_stride / 64);
    const int c_iters = (int)std::max<size_t>(4, ((size_t)4 << 30) / chunk);
    std::vector<double> rr(N), rs(N);
    for (int rep = 0; rep < 2; ++rep) {
        for (size_t k = 0; k < N; ++k) {
            double *b = (double *)((char *)va + k * chunk);
            const double r = run(true, b, c_stride, slots, c_rows, c_iters), s = run(false, b, c_stride, slots, c_rows, c_iters);
            if (rep == 1) printf("chunk %3zu  rnd %.2f (rep0 %.2f)  seq %.2f (rep0 %.2f)\n", k, r, rr[k], s, rs[k]);
            rr[k] = r; rs[k] = s;
        }
    }
    {
        std::vector<double> t = rr; std::sort(t.begin(), t.end());
        printf("per-chunk rnd: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f TB/s\n", t.front(), t[N / 10], t[N / 2], t[N * 9 / 10], t.back());
        t = rs; std::sort(t.begin(), t.end());
        printf("per-chunk seq: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f TB/s\n", t.front(), t[N / 10], t[N / 2], t[N * 9 / 10], t.back());
    }
    fflush(stdout);

    // ---- step 2: C3-geometry workspaces assembled from chosen chunks
    const size_t pad = 1053184;
    const int rows = 65536;
    const size_t stride = (size_t)rows * 64 + pad / 8;                    // doubles
    const size_t ws_bytes = (size_t)slots * stride * 8;
    const size_t K = (ws_bytes + chunk - 1) / chunk;                      // chunks per workspace
    if (K > N) { printf("not enough chunks for a workspace (%zu needed)\n", K); return 0; }
    CK(hipMemUnmap(va, N * chunk));
    void *w1 = nullptr, *w2 = nullptr;
    CK(hipMemAddressReserve(&w1, K * chunk, 0, nullptr, 0));
    CK(hipMemAddressReserve(&w2, (K + K / 2) * chunk, 0, nullptr, 0));
    auto test = [&](const char *name, const std::vector<size_t> &set, void *base) {
        for (size_t q = 0; q < K; ++q) CK(hipMemMap((char *)base + q * chunk, chunk, 0, h[set[q]], 0));
        CK(hipMemSetAccess(base, K * chunk, &acc, 1));
        double r[2], s[2];
        for (int rep = 0; rep < 2; ++rep) {
            r[rep] = run(true, (double *)base, stride, slots, rows, 4);
            s[rep] = run(false, (double *)base, stride, slots, rows, 4);
        }
        double mean_r = 0, mean_s = 0;
        for (size_t q = 0; q < K; ++q) { mean_r += rr[set[q]] / K; mean_s += rs[set[q]] / K; }
        printf("%-28s @%p: rnd %.2f %.2f  seq %.2f %.2f TB/s   (mean per-chunk rnd %.2f seq %.2f)\n", name, base, r[0], r[1], s[0], s[1], mean_r, mean_s);
        fflush(stdout);
        CK(hipMemUnmap(base, K * chunk));
    };
    std::vector<size_t> set(K);
    for (size_t w0 = 0; w0 + K <= N; w0 += K) {
        std::iota(set.begin(), set.end(), w0);
        char nm[64];
        snprintf(nm, sizeof nm, "window chunks %zu..%zu", w0, w0 + K - 1);
        test(nm, set, w1);
    }
    std::vector<size_t> order(N);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return rr[a] + rs[a] > rr[b] + rs[b]; });
    std::vector<size_t> fast(order.begin(), order.begin() + K), slow(order.end() - K, order.end());
    test("fastest chunks", fast, w1);
    test("slowest chunks", slow, w1);
    test("fastest chunks, other VA", fast, (char *)w2 + (K / 2) * chunk);
    test("slowest chunks, other VA", slow, (char *)w2 + (K / 2) * chunk);
    {
        std::vector<size_t> sh = fast;
        unsigned long long sd = 88172645463325252ull;
        for (size_t i = sh.size() - 1; i > 0; --i) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(sh[i], sh[sd % (i + 1)]); }
        test("fastest chunks, shuffled", sh, w1);
        std::vector<size_t> sorted_idx = fast;
        std::sort(sorted_idx.begin(), sorted_idx.end());
        test("fastest chunks, index order", sorted_idx, w1);
        // every other chunk / strided picks: does the SPACING of the physical chunks matter?
        std::vector<size_t> ev;
        for (size_t k = 0; k < N && ev.size() < K; k += 2) ev.push_back(k);
        if (ev.size() == K) test("even chunks 0,2,4,...", ev, w1);
        ev.clear();
        for (size_t k = 0; k < N && ev.size() < K; k += 4) ev.push_back(k);
        if (ev.size() == K) test("chunks 0,4,8,...", ev, w1);
    }
    // compare with what plain hipMalloc gives in this process state
    for (int c = 0; c < 3; ++c) {
        void *q = nullptr;
        if (hipMalloc(&q, ws_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(q, 0, ws_bytes));
        const double r = run(true, (double *)q, stride, slots, rows, 4), s = run(false, (double *)q, stride, slots, rows, 4);
        printf("hipMalloc candidate %d @%p: rnd %.2f seq %.2f TB/s\n", c, q, r, s);
        // (held, so that the next one lands elsewhere)
    }
    return 0;
}
