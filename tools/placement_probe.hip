// placement_probe.hip -- does the physical placement of the message workspace in HBM matter?
// Holds K candidate workspaces of the C3 size (768 slots x 32 MiB) at the same time and runs the
// same two sweeps on each: "seq" = every workgroup streams its slot in place (check-sweep pattern),
// "rnd" = every wave gathers 4 random 512-byte rows of its slot and scatters them back
// (variable-sweep pattern).  Prints TB/s (read+write) per candidate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/placement_probe tools/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(512) rnd_sweep(double *base, size_t slot_stride, const int *__restrict__ perm, int rows, int iters)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * slot_stride + lane;
    for (int it = 0; it < iters; ++it) {
        for (int j = w; j < rows / 4; j += 8) {
            const int *p = perm + 4 * j;
            double c0 = M[(size_t)p[0] * 64], c1 = M[(size_t)p[1] * 64], c2 = M[(size_t)p[2] * 64], c3 = M[(size_t)p[3] * 64];
            M[(size_t)p[0] * 64] = c1 * 1.0000001; M[(size_t)p[1] * 64] = c2 * 1.0000001;
            M[(size_t)p[2] * 64] = c3 * 1.0000001; M[(size_t)p[3] * 64] = c0 * 1.0000001;
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

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 8;
    const int slots = 768, rows = 65536, iters = 6;
    const size_t pad_bytes = argc > 2 ? (size_t)atoll(argv[2]) : 1053184;
    const size_t slot_stride = (size_t)rows * 64 + (pad_bytes / 8);
    const size_t bytes = (size_t)slots * ((size_t)rows * 64 + (20u << 20) / 8) * sizeof(double);   // room for any pad tried
    printf("slot pad %zu bytes\n", pad_bytes);
    std::vector<int> perm(rows);
    for (int i = 0; i < rows; ++i) perm[i] = i;
    unsigned long long s = 88172645463325252ull;
    for (int i = rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int j = (int)(s % (unsigned)(i + 1)); std::swap(perm[i], perm[j]); }
    int *dperm; CK(hipMalloc(&dperm, rows * sizeof(int))); CK(hipMemcpy(dperm, perm.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double *> buf(K);
    for (int k = 0; k < K; ++k) { CK(hipMalloc(&buf[k], bytes)); CK(hipMemset(buf[k], 0, bytes)); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < K; ++k) {
            float ms_r, ms_s;
            hipLaunchKernelGGL(rnd_sweep, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, dperm, rows, 1);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(rnd_sweep, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, dperm, rows, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_r, a, b));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(seq_sweep, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, rows, iters);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms_s, a, b));
            const double tb = 2.0 * (double)slots * rows * 512 * iters / 1e12;
            printf("rep %d candidate %d @%p : rnd %.2f TB/s  seq %.2f TB/s\n", rep, k, (void *)buf[k], tb / (ms_r * 1e-3), tb / (ms_s * 1e-3));
        }
    return 0;
}
