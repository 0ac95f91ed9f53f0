// Can the accumulator registers (AGPRs) of gfx950 be addressed with a run-time index?  (tools; see DESIGN.md "Rows in registers")
//   test 1: v_accvgpr_read_b32 / v_accvgpr_write_b32 under s_set_gpr_idx_on (VGPR indexing mode: SRC0 / DST relative to M0)
//   test 2: the same through a scalar branch tree with static register numbers (always legal; the fallback)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/agpr_index_probe tools/agpr_index_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define AGPR_CLOBBERS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
  "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"

template <int LO, int HI>
__device__ __forceinline__ unsigned tree_get(const int q)
{
    if constexpr (LO == HI) {
        unsigned v;
        asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(v) : "n"(LO));
        return v;
    } else {
        constexpr int MID = (LO + HI) / 2;
        return q <= MID ? tree_get<LO, MID>(q) : tree_get<MID + 1, HI>(q);
    }
}
template <int LO, int HI>
__device__ __forceinline__ void tree_put(const int q, const unsigned v)
{
    if constexpr (LO == HI) {
        asm volatile("v_accvgpr_write_b32 a%1, %0" : : "v"(v), "n"(LO) : AGPR_CLOBBERS);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (q <= MID) tree_put<LO, MID>(q, v);
        else tree_put<MID + 1, HI>(q, v);
    }
}
__device__ __forceinline__ unsigned idx_get(const int q)   // a[q] with the index mode
{
    unsigned v;
    asm volatile("s_set_gpr_idx_on %1, 0x1\n\tv_accvgpr_read_b32 %0, a0\n\ts_set_gpr_idx_off" : "=v"(v) : "s"(q));
    return v;
}
__device__ __forceinline__ void idx_put(const int q, const unsigned v)
{
    asm volatile("s_set_gpr_idx_on %1, 0x8\n\tv_accvgpr_write_b32 a0, %0\n\ts_set_gpr_idx_off" : : "v"(v), "s"(q) : AGPR_CLOBBERS);
}

// out[0..63][lane]: tree-written values read back with the index mode; out[64..127]: index-written values read back by the tree
__global__ void probe(unsigned *out, const int *perm)
{
    const unsigned lane = threadIdx.x;
    for (int q = 0; q < 64; ++q) tree_put<0, 63>(q, 1000u * (unsigned)q + lane);
    for (int q = 0; q < 64; ++q) {
        const int p = __builtin_amdgcn_readfirstlane(perm[q]);
        out[(size_t)q * 64 + lane] = idx_get(p);
    }
    for (int q = 0; q < 64; ++q) {
        const int p = __builtin_amdgcn_readfirstlane(perm[q]);
        idx_put(p, 77000u + 1000u * (unsigned)p + lane);
    }
    for (int q = 0; q < 64; ++q) out[(size_t)(64 + q) * 64 + lane] = tree_get<0, 63>(q);
}

int main()
{
    unsigned *d_out; int *d_perm;
    std::vector<int> perm(64);
    for (int q = 0; q < 64; ++q) perm[q] = (q * 37 + 11) % 64;
    hipMalloc(&d_out, 128 * 64 * 4); hipMalloc(&d_perm, 64 * 4);
    hipMemset(d_out, 0xff, 128 * 64 * 4);
    hipMemcpy(d_perm, perm.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_perm);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 2; }
    std::vector<unsigned> out(128 * 64);
    hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
    int bad_read = 0, bad_write = 0;
    for (int q = 0; q < 64; ++q)
        for (int l = 0; l < 64; ++l) {
            if (out[(size_t)q * 64 + l] != 1000u * (unsigned)perm[q] + l) ++bad_read;
            if (out[(size_t)(64 + q) * 64 + l] != 77000u + 1000u * (unsigned)q + l) ++bad_write;
        }
    std::printf("indexed v_accvgpr_read : %s (%d mismatches; a[perm[1]=%d] read as %u, expected %u)\n", bad_read ? "NOT usable" : "works", bad_read, perm[1], out[64], 1000u * perm[1]);
    std::printf("indexed v_accvgpr_write: %s (%d mismatches; a[1] holds %u, expected %u)\n", bad_write ? "NOT usable" : "works", bad_write, out[65 * 64], 78000u);
    return 0;
}
