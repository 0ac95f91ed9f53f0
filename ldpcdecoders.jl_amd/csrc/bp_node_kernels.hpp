// bp_node_kernels.hpp -- small batches on Tanner graphs too large for the LDS kernel.
//
// The tile kernel (bp_kernels.hpp) puts one syndrome in one LANE: a batch of 1 (a plain decode!)
// or of a few hundred on an n = 16384 code leaves the chip idle and takes 7 ... 136 ms, slower than
// one CPU core.  Here one WORKGROUP owns one syndrome and a work item is one Tanner-graph node, as
// in bp_lds_kernels.hpp with S = 1 -- but the edge messages live in a private slot of a global
// workspace (nnz doubles, CSR order, updated in place), where the workgroup's own CU keeps them in
// its L2: check i reads and writes its deg(i) contiguous messages, bit j gathers / scatters its
// deg(j) messages through csc2csr.  Syndrome and hard-decision bytes sit in LDS; the graph is read
// from global memory (coalesced along the node index).  Same arithmetic, same order as the other
// two kernels and src/decoders/belief_propagation.jl:121-188 (the node updates ARE
// lds_check_unit / lds_bit_unit).
//
// Throughput is poor by design: every message moves as an 8-byte access of its own, and a CU's
// texture-address path retires about one such lane access per clock (n = 16384, nnz = 65536: ~123 us
// per iteration, matching 2 x 65536 gathers/scatters + 65536 64-byte-strided pairs; batching several
// nodes per thread to overlap the index -> message round trips changed nothing, so latency is not the
// bound).  The host only picks this kernel while the tile kernel would leave most CUs without a tile.
#pragma once
#include "bp_lds_kernels.hpp"

namespace ldpc {

struct NodeParams {
    int s, n, nnz;
    int max_iters;
    long long batch;
    double r;
    const unsigned char *syn;   // [batch][s]
    unsigned char *err;         // [batch][n]
    unsigned char *conv;        // [batch]
    int *iters;                 // [batch] or nullptr
    double *llr;                // [batch][n] or nullptr
    int llr_exact;              // LLRs from the full posterior odds (bp_kernels.hpp llr_of)
    double *msg;                // [gridDim.x][slot_stride] workspace
    long long slot_stride;      // doubles per workgroup slot (>= nnz)
    unsigned int *queue;
    u64 *sum_iters;
    // as the last pass of the straggler hand-off (bp_kernels.hpp): syndrome q of this launch is syndrome index[q]
    // of the batch, there are *count_dev of them, and above count_max the tile / team kernels take them.  It has
    // it0[q] iterations behind it, and its messages wait in lane q % 64 of packed tile q / 64 of `state`
    // (tile layout msg[edge][64]): the workgroup gathers that column into its own message array and counts on.
    const int *index;
    const unsigned int *count_dev;
    unsigned int count_max;
    const int *it0;
    const double *state;
    long long state_stride;     // doubles between two packed tiles
    u64 *next_ctrl;             // see LdsParams
    // latency mode (see LdsParams): queue == nullptr, workgroup g decodes syndrome g only
    unsigned int *done_count;
    unsigned int *done_flag;
    unsigned int done_ticket;
    // hybrid placement (MSG == 2): the messages of checks [0, split_check) = edges [0, split_edge) sit in LDS,
    // the rest in the global slot (edge e at msg[e - split_edge])
    int split_check, split_edge;
};

__host__ __device__ inline size_t node_lds_bytes(int s, int n) { return ((size_t)s + (size_t)n + 15) & ~(size_t)15; }

// Variable-node update with the message array split between LDS and the global slot (MSG == 2): lds_bit_unit's
// arithmetic on addresses chosen per edge (generic pointers: the hardware routes each access).
template <int DV>
__device__ __forceinline__ double hybrid_bit_unit(double *Ml, double *Mg, int split, const int *pos, int deg, double r)
{
    double F = r;                                                     // :153
    if (deg <= DV) {
        double c[DV], pre[DV];
        double *at[DV];
#pragma unroll
        for (int k = 0; k < DV; ++k)
            if (k < deg) { const int e = pos[k]; at[k] = e < split ? Ml + e : Mg + (e - split); c[k] = *at[k]; }
#pragma unroll
        for (int k = 0; k < DV; ++k)
            if (k < deg) {
                pre[k] = F;                                           // :156
                F = F * c[k];                                         // :157
                if (F != F) F = 1.0;                                  // :158-160
            }
        double G = 1.0;                                               // :170
#pragma unroll
        for (int k = DV - 1; k >= 0; --k)
            if (k < deg) {
                *at[k] = pre[k] * G;                                  // :172
                G = G * c[k];                                         // :173
                if (G != G) G = 1.0;                                  // :174-176
            }
    } else {
        auto addr = [&](int k) { const int e = pos[k]; return e < split ? Ml + e : Mg + (e - split); };
        for (int k = 0; k < deg; ++k) {
            F = F * *addr(k);
            if (F != F) F = 1.0;
        }
        double G = 1.0;
        for (int k = deg - 1; k >= 0; --k) {
            double Pk = r;
            for (int q = 0; q < k; ++q) {
                Pk = Pk * *addr(q);
                if (Pk != Pk) Pk = 1.0;
            }
            double *a = addr(k);
            const double ck = *a;
            *a = Pk * G;
            G = G * ck;
            if (G != G) G = 1.0;
        }
    }
    return F;
}

// MSG: where the nnz messages of the syndrome live.  0 = the workgroup's global slot (any size; the CU's L2
// keeps it).  1 = LDS, behind the syndrome / decision bytes -- graphs whose messages fit a CU's LDS but whose
// uint16 graph copy and 64-bit masks (the LDS kernel's layout) do not: the scattered 8-byte accesses then
// cost LDS cycles instead of the CU's address path.  2 = hybrid: as many leading checks' messages in LDS as
// fit, the rest in the global slot.
template <int DC, int DV, bool WANT_LLR, int THREADS, int MSG>
__global__ void __launch_bounds__(THREADS)
bp_node_kernel(NodeParams p, const int *__restrict__ row_ptr, const int *__restrict__ edge_bit,
               const int *__restrict__ col_ptr, const int *__restrict__ csc2csr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char node_lds[];
    unsigned char *sbit = node_lds;           // [s] syndrome entry parity
    unsigned char *ebit = node_lds + p.s;     // [n] hard decisions
    __shared__ long long sh_b;
    __shared__ long long sh_q;
    const int s = p.s, n = p.n;
    const int tid = threadIdx.x;
    const double r = p.r;
    double *const Ml = (double *)(node_lds + node_lds_bytes(p.s, p.n));
    double *const Mg = p.msg + (size_t)blockIdx.x * (size_t)p.slot_stride;
    double *const M = MSG == 1 ? Ml : Mg;                      // MSG 0 / 1: the one array
    const int split_check = p.split_check, split_edge = p.split_edge;   // MSG 2
    u64 acc_iters = 0;
    if (p.next_ctrl && blockIdx.x == 0 && tid < 8) p.next_ctrl[tid] = 0;
    long long batch = p.batch;
    if (p.count_dev) {
        batch = (long long)*p.count_dev;
        if (batch > (long long)p.count_max) batch = 0;
    }
    bool taken = false;
    for (;;) {
        long long b;
        if (!p.queue) {                        // latency mode: exactly one syndrome per workgroup
            b = (taken || (long long)blockIdx.x >= batch) ? -1 : (long long)blockIdx.x;
            taken = true;
        } else {
            if (tid == 0) {
                const long long q = (long long)atomicAdd(p.queue, 1u);
                sh_b = q >= batch ? -1 : (p.index ? (long long)p.index[q] : q);
                sh_q = q;
            }
            __syncthreads();
            b = sh_b;
        }
        if (b < 0) break;                      // every wave of every workgroup reaches this
        // ---- a handed-off syndrome: its messages out of the packed tile (one 8-byte word per 512-byte row)
        const bool resumed = p.state != nullptr;
        int it = 0;
        if (resumed) {
            const long long q = sh_q;
            const double *src = p.state + (size_t)(q >> 6) * (size_t)p.state_stride + (size_t)(q & 63);
            for (int e = tid; e < p.nnz; e += THREADS) {
                const double v = src[(size_t)e * kTile];
                if (MSG == 2) { if (e < split_edge) Ml[e] = v; else Mg[e - split_edge] = v; }
                else M[e] = v;
            }
            it = p.it0[q];
        }
        // ---- syndrome in (:136: entries > 1 can never be matched by a parity bit)
        int bad = 0;
        for (int i = tid; i < s; i += THREADS) {
            const unsigned v = p.syn[(size_t)b * s + i];
            sbit[i] = (unsigned char)(v & 1u);
            bad |= (v > 1u);
        }
        const int never = __syncthreads_or(bad);   // also orders sbit[] before the sweeps

        int converged = 0;
        while (it < p.max_iters) {
            ++it;
            const bool first = (it == 1) && !resumed;
            // ---- check sweep: one thread per check, its messages are contiguous
            for (int i = tid; i < s; i += THREADS) {
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const double sigma = sbit[i] ? -1.0 : 1.0;
                if (MSG == 2) {
                    if (i < split_check) lds_check_unit<DC>(Ml + e0, 1, deg, sigma, first, r);
                    else lds_check_unit<DC>(Mg + (e0 - split_edge), 1, deg, sigma, first, r);
                } else {
                    lds_check_unit<DC>(M + e0, 1, deg, sigma, first, r);
                }
            }
            __syncthreads();
            // ---- variable sweep: one thread per bit
            for (int j = tid; j < n; j += THREADS) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                const double T = MSG == 2 ? hybrid_bit_unit<DV>(Ml, Mg, split_edge, csc2csr + c0, deg, r)
                                          : lds_bit_unit<DV, int>(M, 1, csc2csr + c0, deg, r);
                if (WANT_LLR) p.llr[(size_t)b * n + j] = llr_of(T, p.llr_exact);   // :163, final at the last iteration run
                ebit[j] = (unsigned char)(T >= 1.0);                      // :164-168
            }
            __syncthreads();
            // ---- convergence test (:180-184)
            int mism = never;
            for (int i = tid; i < s; i += THREADS) {
                unsigned par = sbit[i];
                const int e1 = row_ptr[i + 1];
                for (int e = row_ptr[i]; e < e1; ++e) par ^= ebit[edge_bit[e]];
                mism |= (int)par;
            }
            if (!__syncthreads_or(mism)) { converged = 1; break; }
        }
        // ---- results out
        for (int j = tid; j < n; j += THREADS) p.err[(size_t)b * n + j] = (p.max_iters > 0) ? ebit[j] : (unsigned char)0;
        if (tid == 0) {
            p.conv[b] = (unsigned char)converged;
            if (p.iters) p.iters[b] = it;
            acc_iters += (u64)it;
        }
        __syncthreads();   // ebit[] / sbit[] / sh_b are rewritten by the next syndrome
    }
    if (p.done_flag) publish_done(p.done_count, p.done_flag, p.done_ticket);
    if (tid == 0 && acc_iters && p.sum_iters) atomicAdd(p.sum_iters, acc_iters);
}

}  // namespace ldpc
