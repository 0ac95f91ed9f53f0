// vmm_probe2.hip -- follow-up of vmm_probe.hip, which showed that the speed of the sweeps over a C3-size
// workspace does NOT depend on which physical chunks back it (any 25 of 150 chunks, any order: 5.88 / 5.68 TB/s)
// but DOES change when the very same chunks are mapped at another virtual address (5.61 / 5.53).
// Here: ONE set of physical chunks, mapped at many virtual bases inside one huge reservation --
//   * base = R + j * 2 MiB, 32 MiB, 1 GiB, 64 GiB  (which address bits matter?)
//   * chunk size 2 MiB / 64 MiB / 1 GiB at the best and worst base (does physical contiguity matter?)
//   * slot padding 0 / 4.5 KiB / 1 MiB + 4.5 KiB / 2 MiB + 4.5 KiB at the best and worst base
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe2 tools/vmm_probe2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
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
static const int slots = 768, rows = 65536;

static double run(bool rnd, double *base, size_t stride_doubles, int iters)
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

struct Phys {
    size_t chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
    size_t bytes() const { return chunk * h.size(); }
};

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;

static Phys make_phys(size_t total, size_t chunk)
{
    Phys p;
    p.chunk = chunk;
    for (size_t k = 0; k * chunk < total; ++k) {
        hipMemGenericAllocationHandle_t q;
        CK(hipMemCreate(&q, chunk, &prop, 0));
        p.h.push_back(q);
    }
    return p;
}
static void free_phys(Phys &p)
{
    for (auto q : p.h) CK(hipMemRelease(q));
    p.h.clear();
}
static void map_at(const Phys &p, char *base)
{
    for (size_t k = 0; k < p.h.size(); ++k) CK(hipMemMap(base + k * p.chunk, p.chunk, 0, p.h[k], 0));
    CK(hipMemSetAccess(base, p.bytes(), &acc, 1));
}

int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    const size_t pad0 = 1053184;
    const size_t ws_max = (size_t)slots * ((size_t)rows * 512 + 3 * MiB);    // room for every pad tried
    const size_t total = (ws_max + GiB - 1) / GiB * GiB;
    // one huge reservation, 1 GiB aligned; every base below is R + something
    const size_t resv = (size_t)1200 * GiB;
    void *Rv = nullptr;
    if (hipMemAddressReserve(&Rv, resv, GiB, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); CK(hipMemAddressReserve(&Rv, resv, 0, nullptr, 0)); }
    char *R = (char *)Rv;
    printf("reservation of %zu GiB at %p (asked for 1 GiB alignment)\n", resv / GiB, Rv);
    R = (char *)(((uintptr_t)R + GiB - 1) & ~(uintptr_t)(GiB - 1));
    printf("aligned base R = %p\n", (void *)R);

    Phys big = make_phys(total, GiB);
    map_at(big, R);
    CK(hipMemset(R, 0, big.bytes()));
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(R, big.bytes()));

    struct Res { std::string name; double r, s; };
    std::vector<Res> all;
    auto test = [&](const Phys &p, const std::string &name, size_t off, size_t pad) {
        char *base = R + off;
        map_at(p, base);
        const size_t stride = ((size_t)rows * 512 + pad) / 8;
        double r = 0, s = 0;
        for (int rep = 0; rep < 2; ++rep) { r = std::max(r, run(true, (double *)base, stride, 4)); s = std::max(s, run(false, (double *)base, stride, 4)); }
        printf("%-44s base %p pad %8zu: rnd %.2f seq %.2f TB/s\n", name.c_str(), (void *)base, pad, r, s);
        fflush(stdout);
        CK(hipMemUnmap(base, p.bytes()));
        all.push_back({name, r, s});
        return r + s;
    };
    double best = 0, worst = 1e9;
    size_t best_off = 0, worst_off = 0;
    auto track = [&](double v, size_t off) { if (v > best) { best = v; best_off = off; } if (v < worst) { worst = v; worst_off = off; } };
    char nm[96];
    track(test(big, "R", 0, pad0), 0);
    for (size_t j = 1; j <= 16; ++j) { snprintf(nm, sizeof nm, "R + %zu x 2 MiB", j); track(test(big, nm, j * 2 * MiB, pad0), j * 2 * MiB); }
    for (size_t j = 2; j <= 16; ++j) { snprintf(nm, sizeof nm, "R + %zu x 32 MiB", j); track(test(big, nm, j * 32 * MiB, pad0), j * 32 * MiB); }
    for (size_t j = 1; j <= 16; ++j) { snprintf(nm, sizeof nm, "R + %zu GiB", j); track(test(big, nm, j * GiB, pad0), j * GiB); }
    for (size_t j = 1; j <= 16; ++j) { snprintf(nm, sizeof nm, "R + %zu x 64 GiB", j); track(test(big, nm, j * 64 * GiB, pad0), j * 64 * GiB); }
    for (size_t j = 1; j <= 8; ++j) { snprintf(nm, sizeof nm, "R + %zu x 4 KiB", j); track(test(big, nm, j * 4096, pad0), j * 4096); }
    printf("best offset %zu (%.2f), worst offset %zu (%.2f)\n", best_off, best, worst_off, worst);

    // padding at the best and the worst base
    for (size_t off : {best_off, worst_off})
        for (size_t pad : {(size_t)0, (size_t)4608, (size_t)65536 + 4608, pad0, 2 * MiB + 4608, 3 * MiB - 512}) {
            snprintf(nm, sizeof nm, "pad sweep at offset %zu", off);
            test(big, nm, off, pad);
        }
    free_phys(big);
    // physical chunk size at the best and the worst base
    for (size_t chunk : {2 * MiB, 64 * MiB}) {
        Phys p = make_phys(total, chunk);
        map_at(p, R);
        CK(hipMemset(R, 0, p.bytes()));
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(R, p.bytes()));
        for (size_t off : {best_off, worst_off}) {
            snprintf(nm, sizeof nm, "chunks of %zu MiB at offset %zu", chunk / MiB, off);
            test(p, nm, off, pad0);
        }
        free_phys(p);
    }
    return 0;
}
