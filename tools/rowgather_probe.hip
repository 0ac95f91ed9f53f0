// rowgather_probe.hip -- does the variable sweep's access pattern (gather 4 random rows, scatter them
// back, in place) run faster with 1 KiB rows (16 B per lane) than with 512 B rows (8 B per lane)?
// Build: hipcc --offload-arch=gfx950 -O3 -o rowgather_probe tools/rowgather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one workgroup per tile slot; rows_per_tile rows of ROWB bytes; every wave visits "bits": 4 random rows each
template <typename T>
__global__ void __launch_bounds__(512) sweep(T *base, const int *__restrict__ perm, int rows_per_tile, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T *M = base + (size_t)blockIdx.x * rows_per_tile * 64 + lane;
    for (int it = 0; it < iters; ++it) {
        for (int j = w; j < rows_per_tile / 4; j += 8) {
            const int *p = perm + 4 * j;
            T c0 = M[(size_t)p[0] * 64], c1 = M[(size_t)p[1] * 64], c2 = M[(size_t)p[2] * 64], c3 = M[(size_t)p[3] * 64];
            M[(size_t)p[0] * 64] = c1 * 1.0000001; M[(size_t)p[1] * 64] = c2 * 1.0000001;
            M[(size_t)p[2] * 64] = c3 * 1.0000001; M[(size_t)p[3] * 64] = c0 * 1.0000001;
        }
        __syncthreads();
    }
}

typedef double d2 __attribute__((ext_vector_type(2)));

template <typename T>
double run(int slots, int rows, int iters)
{
    T *buf; int *dperm;
    size_t bytes = (size_t)slots * rows * 64 * sizeof(T);
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    std::vector<int> perm(rows);
    for (int i = 0; i < rows; ++i) perm[i] = i;
    unsigned long long s = 88172645463325252ull;
    for (int i = rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int j = (int)(s % (unsigned)(i + 1)); std::swap(perm[i], perm[j]); }
    CK(hipMalloc(&dperm, rows * sizeof(int))); CK(hipMemcpy(dperm, perm.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(sweep<T>, dim3(slots), dim3(512), 0, 0, buf, dperm, rows, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(sweep<T>, dim3(slots), dim3(512), 0, 0, buf, dperm, rows, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipFree(buf)); CK(hipFree(dperm));
    return 2.0 * bytes * iters / (ms * 1e-3) / 1e12;
}

int main()
{
    // same bytes per slot (32 MiB) and the same 768 slots as the C3 workspace
    printf("512 B rows (8 B/lane) : %.2f TB/s (r+w)\n", run<double>(768, 65536, 10));
    printf("1 KiB rows (16 B/lane): %.2f TB/s (r+w)\n", run<d2>(768, 32768, 10));
    printf("512 B rows, 384 slots : %.2f TB/s (r+w)\n", run<double>(384, 65536, 10));
    printf("1 KiB rows, 384 slots : %.2f TB/s (r+w)\n", run<d2>(384, 32768, 10));
    return 0;
}
