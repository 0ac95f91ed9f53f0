// vmm_probe5.hip -- is the slow / fast placement of the workspace really about WHICH WORKGROUP (hence which XCD:
// workgroups are dealt round-robin over the 8 XCDs) sweeps WHICH physical region?  vmm_probe4: physically contiguous
// workspaces (hipMalloc, one hipMemCreate) run 5.0-5.5 TB/s, the same memory cut into 64 MiB chunks and mapped in
// shuffled order 6.04 TB/s.  If so, no virtual-memory games are needed: a permutation of the slot index does it.
// One plain hipMalloc; the kernels take a table  slot_of[blockIdx]:
//   identity | xcd-major (XCD x gets slots [96x, 96x+96)) | (b * m) mod 768 for a few multipliers | random permutations
// Usage: tools/vmm_probe5 [hold_GiB]      Build: hipcc --offload-arch=gfx950 -O3 -o tools/vmm_probe5 tools/vmm_probe5.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) var_like(double *base, size_t slot_stride, int rows, int iters, const int *__restrict__ slot_of)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)slot_of[blockIdx.x] * slot_stride + lane;
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

__global__ void __launch_bounds__(512) check_like(double *base, size_t slot_stride, int rows, int iters, int rotate, const int *__restrict__ slot_of)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)slot_of[blockIdx.x] * slot_stride + lane;
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

static double run(int kind, double *base, size_t stride_doubles, int iters, const int *tab)
{
    auto launch = [&](int n) {
        if (kind == 0) hipLaunchKernelGGL(var_like, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, n, tab);
        else hipLaunchKernelGGL(check_like, dim3(slots), dim3(512), 0, 0, base, stride_doubles, rows, n, 1, tab);
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

int main(int argc, char **argv)
{
    const size_t hold_gib = argc > 1 ? (size_t)atoll(argv[1]) : 0;
    CK(hipSetDevice(0));
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t pad = 1053184;
    const size_t stride = ((size_t)rows * 512 + pad) / 8;
    void *hold = nullptr;
    if (hold_gib) CK(hipMalloc(&hold, hold_gib << 30));
    const size_t ws = (size_t)slots * stride * 8;
    double *base = nullptr;
    CK(hipMalloc((void **)&base, ws));
    CK(hipMemset(base, 0, ws));
    int *tab = nullptr;
    CK(hipMalloc((void **)&tab, slots * sizeof(int)));
    printf("hipMalloc workspace @%p (hold %zu GiB)\n", (void *)base, hold_gib);
    auto test = [&](const std::string &name, const std::vector<int> &t) {
        std::vector<int> chk = t; std::sort(chk.begin(), chk.end());
        for (int i = 0; i < slots; ++i) if (chk[i] != i) { printf("%s: not a permutation\n", name.c_str()); return; }
        CK(hipMemcpy(tab, t.data(), slots * sizeof(int), hipMemcpyHostToDevice));
        double v = 0, c = 0;
        for (int rep = 0; rep < 2; ++rep) { v = std::max(v, run(0, base, stride, 3, tab)); c = std::max(c, run(1, base, stride, 3, tab)); }
        printf("%-34s var %.2f  check(rot) %.2f TB/s\n", name.c_str(), v, c);
        fflush(stdout);
    };
    std::vector<int> t(slots);
    for (int b = 0; b < slots; ++b) t[b] = b;
    test("identity", t);
    for (int b = 0; b < slots; ++b) t[b] = (b % 8) * (slots / 8) + b / 8;
    test("xcd-major (XCD x: slots 96x..96x+95)", t);
    for (int b = 0; b < slots; ++b) t[b] = (b / 8) + (7 - b % 8) * (slots / 8);
    test("xcd-major reversed", t);
    for (int m : {5, 7, 11, 13, 37, 101, 331, 385, 769 % 768 + 6}) {
        for (int b = 0; b < slots; ++b) t[b] = (int)(((long long)b * m) % slots);
        test("(b * " + std::to_string(m) + ") mod 768", t);
    }
    // keep b % 8 (the XCD) but shuffle which slots of that residue class... and the opposite: rotate the residue
    for (int sh = 1; sh < 8; ++sh) {
        for (int b = 0; b < slots; ++b) t[b] = (b / 8) * 8 + (b % 8 + sh * (b / 8)) % 8;
        test("XCD residue rotated by " + std::to_string(sh) + " per group of 8", t);
    }
    unsigned long long sd = 88172645463325252ull;
    for (int trial = 0; trial < 6; ++trial) {
        for (int b = 0; b < slots; ++b) t[b] = b;
        for (int i = slots - 1; i > 0; --i) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; std::swap(t[i], t[sd % (unsigned)(i + 1)]); }
        test("random permutation " + std::to_string(trial), t);
    }
    for (int b = 0; b < slots; ++b) t[b] = b;
    test("identity again", t);
    return 0;
}
