// interleave_probe.hip -- would a ROW-major interleaving of the workspace (row r of all slots adjacent:
// msg[row][slot][64]) beat the slot-major layout (msg[slot][row][64]) that the tile kernel uses?  With
// every workgroup walking the same graph at about the same pace, the interleaved layout turns the variable
// sweep's "768 scattered 512-byte rows" into one dense 384 KiB band at a time.  Same two sweeps as
// placement_probe.hip (seq = check-sweep pattern, rnd = variable-sweep pattern: gather 4 rows, scatter
// them back), both layouts on the same allocation, plus the interleaved layout with workgroups started
// at random phases (what a persistent kernel looks like after its first generation of tiles).
// Build: hipcc --offload-arch=gfx950 -O3 -o interleave_probe tools/interleave_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// address of (slot, row): slot-major or row-major
template <bool ROWMAJOR>
__device__ __forceinline__ double *at(double *base, size_t slot, size_t row, size_t rows, size_t slots, size_t slot_stride)
{
    return ROWMAJOR ? base + (row * slots + slot) * 64 : base + slot * slot_stride + row * 64;
}

template <bool ROWMAJOR>
__global__ void __launch_bounds__(512) rnd_sweep(double *base, size_t slot_stride, const int *__restrict__ perm, int rows, int iters, int phase_mode)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t slot = blockIdx.x, slots = gridDim.x;
    const int nb = rows / 4;
    const int start = phase_mode ? (int)((blockIdx.x * 2654435761u) % (unsigned)nb) : 0;
    for (int it = 0; it < iters; ++it) {
        for (int j0 = w; j0 < nb; j0 += 8) {
            int j = j0 + start; if (j >= nb) j -= nb;
            const int *p = perm + 4 * j;
            double *a0 = at<ROWMAJOR>(base, slot, p[0], rows, slots, slot_stride) + lane, *a1 = at<ROWMAJOR>(base, slot, p[1], rows, slots, slot_stride) + lane,
                   *a2 = at<ROWMAJOR>(base, slot, p[2], rows, slots, slot_stride) + lane, *a3 = at<ROWMAJOR>(base, slot, p[3], rows, slots, slot_stride) + lane;
            double c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
            *a0 = c1 * 1.0000001; *a1 = c2 * 1.0000001; *a2 = c3 * 1.0000001; *a3 = c0 * 1.0000001;
        }
        __syncthreads();
    }
}

template <bool ROWMAJOR>
__global__ void __launch_bounds__(512) seq_sweep(double *base, size_t slot_stride, int rows, int iters, int phase_mode)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t slot = blockIdx.x, slots = gridDim.x;
    const int nb = rows / 8;
    const int start = phase_mode ? (int)((blockIdx.x * 2246822519u) % (unsigned)nb) : 0;
    for (int it = 0; it < iters; ++it) {
        for (int i0 = w; i0 < nb; i0 += 8) {
            int i = i0 + start; if (i >= nb) i -= nb;
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *(at<ROWMAJOR>(base, slot, (size_t)i * 8 + k, rows, slots, slot_stride) + lane);
#pragma unroll
            for (int k = 0; k < 8; ++k) *(at<ROWMAJOR>(base, slot, (size_t)i * 8 + k, rows, slots, slot_stride) + lane) = v[k] * 1.0000001;
        }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 4;
    const int slots = 768, rows = 65536, iters = 6;
    const size_t pad_bytes = 1053184;
    const size_t slot_stride = (size_t)rows * 64 + pad_bytes / 8;
    const size_t bytes = (size_t)slots * slot_stride * sizeof(double);
    std::vector<int> perm(rows);
    for (int i = 0; i < rows; ++i) perm[i] = i;
    unsigned long long s = 88172645463325252ull;
    for (int i = rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; int j = (int)(s % (unsigned)(i + 1)); std::swap(perm[i], perm[j]); }
    int *dperm; CK(hipMalloc(&dperm, rows * sizeof(int))); CK(hipMemcpy(dperm, perm.data(), rows * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double *> buf(K);
    for (int k = 0; k < K; ++k) { CK(hipMalloc(&buf[k], bytes)); CK(hipMemset(buf[k], 0, bytes)); }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double tb = 2.0 * (double)slots * rows * 512 * iters / 1e12;
    auto timeit = [&](auto launch) { float ms; launch(1); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); launch(iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return tb / (ms * 1e-3); };
    for (int k = 0; k < K; ++k) {
        double r[6];
        r[0] = timeit([&](int n) { hipLaunchKernelGGL(seq_sweep<false>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, rows, n, 1); });
        r[1] = timeit([&](int n) { hipLaunchKernelGGL(rnd_sweep<false>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, dperm, rows, n, 1); });
        r[2] = timeit([&](int n) { hipLaunchKernelGGL(seq_sweep<true>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, rows, n, 0); });
        r[3] = timeit([&](int n) { hipLaunchKernelGGL(rnd_sweep<true>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, dperm, rows, n, 0); });
        r[4] = timeit([&](int n) { hipLaunchKernelGGL(seq_sweep<true>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, rows, n, 1); });
        r[5] = timeit([&](int n) { hipLaunchKernelGGL(rnd_sweep<true>, dim3(slots), dim3(512), 0, 0, buf[k], slot_stride, dperm, rows, n, 1); });
        printf("candidate %d: slot-major (rotated) seq %.2f rnd %.2f | row-major lockstep seq %.2f rnd %.2f | row-major random phases seq %.2f rnd %.2f  TB/s\n",
               k, r[0], r[1], r[2], r[3], r[4], r[5]);
    }
    return 0;
}
