// chunk_probe.hip -- is "fast HBM" a property of individual physical chunks?  Allocates N chunks of
// CH MiB (separate hipMallocs, all held), runs the variable-sweep-like probe on each chunk alone
// (768 workgroups, each owning 1/768 of the chunk) and prints the distribution of TB/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/chunk_probe tools/chunk_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) probe(double *base, long long slot_stride, int rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * (size_t)slot_stride + lane;
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

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 200;
    const size_t chunk_mib = argc > 2 ? (size_t)atoi(argv[2]) : 1024;
    const size_t bytes = chunk_mib << 20;
    const int slots = 768;
    const long long stride = (long long)(bytes / slots / 512) * 64;   // doubles per slot, whole rows
    const int rows = (int)(stride / 64);
    std::vector<double *> buf;
    for (int k = 0; k < N; ++k) { double *q; if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; } CK(hipMemset(q, 0, bytes)); buf.push_back(q); }
    printf("%zu chunks of %zu MiB held, %d rows of 512 B per workgroup\n", buf.size(), chunk_mib, rows);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 20;
    std::vector<double> tb(buf.size());
    for (int rep = 0; rep < 2; ++rep)
        for (size_t k = 0; k < buf.size(); ++k) {
            hipLaunchKernelGGL(probe, dim3(slots), dim3(512), 0, 0, buf[k], stride, rows, 2);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(probe, dim3(slots), dim3(512), 0, 0, buf[k], stride, rows, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            tb[k] = 2.0 * (double)slots * rows * 512 * iters / (ms * 1e-3) / 1e12;
            if (rep == 1) printf("%5.2f%s", tb[k], (k % 16 == 15) ? "\n" : " ");
        }
    printf("\n");
    std::vector<double> s = tb; std::sort(s.begin(), s.end());
    printf("min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f TB/s\n", s.front(), s[s.size() / 10], s[s.size() / 2], s[s.size() * 9 / 10], s.back());
    return 0;
}
