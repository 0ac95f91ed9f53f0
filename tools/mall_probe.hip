// mall_probe.hip -- how fast is the in-place read-modify-write of the message rows when the working set fits the
// 256 MiB Infinity Cache?  The tile kernel keeps 768 tiles x 32 MiB in flight (24 GiB: every sweep is HBM traffic,
// 5.9 TB/s measured = the memory system's ceiling for the mix).  If a working set of 4-6 tiles swept by the WHOLE chip
// runs well above that, a cache-blocked schedule (few tiles at a time, every workgroup on them) beats the ceiling.
//   check-like: a wave owns 8 consecutive 512-B rows: 8 loads, 8 stores in place;
//   var-like:   a wave owns 4 scattered rows.
// No barriers: pure rate.  Waves are re-dealt every pass so no wave re-reads its own lines from L2.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ void __launch_bounds__(512) check_like(double *base, unsigned nnodes, int passes)
{
    const int lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nw = gridDim.x * 8u;
    for (int p = 0; p < passes; ++p) {
        const unsigned gid = (blockIdx.x * 8u + w + (unsigned)p * 2477u) % nw;
        for (unsigned i = gid; i < nnodes; i += nw) {
            double *R = base + (size_t)i * 8 * 64 + lane;
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = R[k * 64];
#pragma unroll
            for (int k = 0; k < 8; ++k) R[k * 64] = v[k] * 1.0000001;
        }
    }
}

__global__ void __launch_bounds__(512) var_like(double *base, unsigned rows, int passes)
{
    const int lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nw = gridDim.x * 8u, nb = rows / 4u;
    for (int p = 0; p < passes; ++p) {
        const unsigned gid = (blockIdx.x * 8u + w + (unsigned)p * 2477u) % nw;
        for (unsigned j = gid; j < nb; j += nw) {
            // four rows of one "bit": a fixed permutation of the row space (bijective: odd multiplier mod 2^k rows)
            const unsigned a = (j * 4u + 0u) * 2654435761u % rows, b = (j * 4u + 1u) * 2654435761u % rows,
                           c = (j * 4u + 2u) * 2654435761u % rows, d = (j * 4u + 3u) * 2654435761u % rows;
            double *M = base + lane;
            const double v0 = M[(size_t)a * 64], v1 = M[(size_t)b * 64], v2 = M[(size_t)c * 64], v3 = M[(size_t)d * 64];
            M[(size_t)a * 64] = v0 * 1.0000001; M[(size_t)b * 64] = v1 * 1.0000001;
            M[(size_t)c * 64] = v2 * 1.0000001; M[(size_t)d * 64] = v3 * 1.0000001;
        }
    }
}

// Teams' pattern: region q (one message slot) is swept only by the workgroups of XCD q (blocks b with b % 8 == q,
// round-robin placement), R regions `stride` bytes apart.  Nodes from nt_from on are accessed with non-temporal
// loads and stores (do they stay out of the Infinity Cache, leaving it to the others?).
__global__ void __launch_bounds__(512) check_like_regions(double *base, size_t stride_doubles, int regions, unsigned nnodes, unsigned nt_from, int passes)
{
    const int lane = threadIdx.x & 63;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned q = blockIdx.x & 7u, m = blockIdx.x >> 3, members = gridDim.x >> 3;
    if ((int)q >= regions) return;
    double *B = base + (size_t)q * stride_doubles;
    const unsigned nw = members * 8u;
    for (int p = 0; p < passes; ++p) {
        const unsigned gid = (m * 8u + w + (unsigned)p * 2477u) % nw;
        for (unsigned i = gid; i < nnodes; i += nw) {
            double *R = B + (size_t)i * 8 * 64 + lane;
            double v[8];
            if (i >= nt_from) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(R + k * 64);
#pragma unroll
                for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k] * 1.0000001, R + k * 64);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = R[k * 64];
#pragma unroll
                for (int k = 0; k < 8; ++k) R[k * 64] = v[k] * 1.0000001;
            }
        }
    }
}

static double run_regions(double *buf, int grid, size_t region_bytes, size_t stride_bytes, int regions, int nt_eighths, hipEvent_t ea, hipEvent_t eb)
{
    const unsigned nnodes = (unsigned)(region_bytes / 512 / 8);
    const unsigned nt_from = nnodes - (unsigned)((size_t)nnodes * nt_eighths / 8);
    const int passes = 300;
    double best = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(check_like_regions, dim3(grid), dim3(512), 0, 0, buf, stride_bytes / 8, regions, nnodes, nt_from, 2);
        CK(hipEventRecord(ea));
        hipLaunchKernelGGL(check_like_regions, dim3(grid), dim3(512), 0, 0, buf, stride_bytes / 8, regions, nnodes, nt_from, passes);
        CK(hipEventRecord(eb));
        CK(hipEventSynchronize(eb));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, ea, eb));
        const double tbs = 2.0 * (double)region_bytes * regions * passes / (ms * 1e-3) / 1e12;
        if (tbs > best) best = tbs;
    }
    return best;
}

static void regions_main(int grid)
{
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t MiB = (size_t)1 << 20, KiB = 1024;
    double *buf = nullptr;
    CK(hipMalloc(&buf, 8 * 48 * MiB));
    CK(hipMemset(buf, 0, 8 * 48 * MiB));
    CK(hipDeviceSynchronize());
    // (1) 8 regions of 32 MiB: which distance between them suits the cache?
    for (size_t pad_kib : {0, 4, 64, 256, 512, 1024, 1028, 1536, 2048, 3072, 4096, 6144, 8192}) {
        const double t = run_regions(buf, grid, 32 * MiB, 32 * MiB + pad_kib * KiB, 8, 0, ea, eb);
        printf("8 regions x 32 MiB, %5zu KiB apart beyond their size, grid %d: %.2f TB/s\n", pad_kib, grid, t);
        fflush(stdout);
    }
    // (2) fewer regions
    for (int regions = 4; regions <= 8; ++regions) {
        const double t = run_regions(buf, grid, 32 * MiB, 33 * MiB, regions, 0, ea, eb);
        printf("%d regions x 32 MiB at 33 MiB stride: %.2f TB/s in all = %.2f per XCD\n", regions, t, t / regions);
        fflush(stdout);
    }
    // (3) beyond the cache: 8 regions of 33 ... 40 MiB, with 0, 1 or 2 eighths of every region accessed non-temporally
    for (size_t reg_mib : {33, 34, 36, 40})
        for (int nt : {0, 1, 2}) {
            const double t = run_regions(buf, grid, reg_mib * MiB, (reg_mib + 1) * MiB, 8, nt, ea, eb);
            printf("8 regions x %zu MiB (%zu MiB in all), %d/8 of each non-temporal: %.2f TB/s\n", reg_mib, 8 * reg_mib, nt, t);
            fflush(stdout);
        }
}

int main(int argc, char **argv)
{
    if (argc > 2 && atoi(argv[2]) == 1) { CK(hipSetDevice(0)); regions_main(atoi(argv[1])); return 0; }
    const int grid = argc > 1 ? atoi(argv[1]) : 768;
    CK(hipSetDevice(0));
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea));
    CK(hipEventCreate(&eb));
    const size_t MiB = (size_t)1 << 20;
    const size_t sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096, 16384};
    double *buf = nullptr;
    CK(hipMalloc(&buf, 16384 * MiB));
    CK(hipMemset(buf, 0, 16384 * MiB));
    CK(hipDeviceSynchronize());
    for (size_t szm : sizes) {
        const size_t W = szm * MiB;
        const unsigned rows = (unsigned)(W / 512), nnodes = rows / 8;
        // about 1 TB of traffic per measurement at least 8 passes
        int passes = (int)(((size_t)400 << 30) / (2 * W));
        if (passes < 4) passes = 4;
        if (passes > 2000) passes = 2000;
        double best[2] = {0, 0};
        for (int kind = 0; kind < 2; ++kind)
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(check_like, dim3(grid), dim3(512), 0, 0, buf, nnodes, 2);
                else hipLaunchKernelGGL(var_like, dim3(grid), dim3(512), 0, 0, buf, rows, 2);
                CK(hipEventRecord(ea));
                if (kind == 0) hipLaunchKernelGGL(check_like, dim3(grid), dim3(512), 0, 0, buf, nnodes, passes);
                else hipLaunchKernelGGL(var_like, dim3(grid), dim3(512), 0, 0, buf, rows, passes);
                CK(hipEventRecord(eb));
                CK(hipEventSynchronize(eb));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, ea, eb));
                const double tbs = 2.0 * (double)W * passes / (ms * 1e-3) / 1e12;
                if (tbs > best[kind]) best[kind] = tbs;
            }
        printf("working set %6zu MiB  grid %d  passes %4d : check-like %.2f TB/s   var-like %.2f TB/s (read+write)\n",
               szm, grid, passes, best[0], best[1]);
        fflush(stdout);
    }
    return 0;
}
