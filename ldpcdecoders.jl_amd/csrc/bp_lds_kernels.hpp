// bp_lds_kernels.hpp -- on-chip variant of the BP sweeps for Tanner graphs whose edge
// messages fit the 160 KiB LDS of one CU (BASELINE configs[0]/[1]: n=1008, nnz=3024;
// configs[4]: BB-72, nnz=216).  Same arithmetic, same order as bp_kernels.hpp and
// src/decoders/belief_propagation.jl:121-188 -- only the placement differs:
//
//   * a workgroup keeps ALL edge messages of S syndromes (S = 1..64, a power of two) in LDS,
//     msg[edge][S] fp64, updated in place, and iterates on them with workgroup barriers only:
//     HBM sees the syndromes once on the way in and the hard decisions once on the way out;
//   * a work item is one (node, syndrome) pair; consecutive lanes hold the S syndromes of one
//     node (conflict-free 8-byte LDS accesses inside a node), consecutive lane groups hold
//     consecutive nodes, so a wave covers 64/S nodes and small batches still fill the machine
//     (a single decode! spreads its 504 checks over 504 lanes instead of one lane);
//   * hard decisions / syndromes are S-bit masks per node in LDS, built with wave ballots;
//   * the Tanner graph (CSR/CSC pointers, edge permutation) is copied into LDS once per
//     persistent workgroup: every index read in the sweeps is a ~64-cycle LDS read instead of a
//     dependent ~1 us L2 round trip (measured: 2.9x on n=1008).
#pragma once
#include "bp_kernels.hpp"
#include "latency_mode.hpp"

#ifndef LDPC_LDS_STAMPS  // 1 = diagnostic build: in-kernel phase stamps (check / variable / rest)
#define LDPC_LDS_STAMPS 0
#endif
#if LDPC_LDS_STAMPS
#define LDS_CLOCK() wall_clock64()
#else
#define LDS_CLOCK() 0ull
#endif

namespace ldpc {

struct LdsParams {
    int s, n, nnz;
    int max_iters;
    int logS;             // S = 1 << logS syndromes per workgroup pass
    int ngroups;          // ceil(batch / S)
    int chunk;            // groups taken from the queue per dequeue (one same-address atomic costs ~11 ns
                          // chip-wide: a dequeue per group would cap the kernel at ~88 groups/us)
    long long batch;
    double r;
    const unsigned char *syn;   // [batch][s]
    unsigned char *err;         // [batch][n]
    unsigned char *conv;        // [batch]
    int *iters;                 // [batch] or nullptr
    double *llr;                // [batch][n] or nullptr
    int llr_exact;              // LLRs from the full posterior odds (bp_kernels.hpp llr_of)
    unsigned int *queue;
    u64 *sum_iters;
    u64 *phase_ticks;     // [3] 100 MHz ticks: check sweep, variable sweep, everything else (I/O, test, barriers)
    // Latency mode of the host-pointer entry (a plain decode!): queue == nullptr, workgroup g decodes
    // group g only (gridDim.x == ngroups); syn / err / ... are host-mapped, and the last workgroup to
    // finish publishes done_ticket in the host-mapped word done_flag, on which the host spins.
    u64 *next_ctrl;             // the NEXT call's 64-byte control slot: zeroed here, so that that call needs no memset
                                // (calls on one handle are ordered; nullptr = leave it alone)
    unsigned int *done_count;   // device word, zero between launches
    unsigned int *done_flag;    // nullptr = nobody is waiting
    unsigned int done_ticket;
};

// Iteration 1 of the check sweep (belief_propagation.jl:135-150 with every bit -> check message still r, :129) only
// depends on the check's degree and on the sign of its syndrome entry: the messages it writes are tabulated once per
// workgroup -- by lds_check_unit itself, so bit for bit what every work item would compute -- and the sweep copies them:
// no fp64 division in the iteration that 90-99 % of the syndromes of the low-error-rate configurations (C1/C2: per 0.01,
// C5: per 0.005) need.  Degrees up to 16 (2 x 17 x 16 doubles); wider checks compute as before.
constexpr int kFirstTabDeg = 16;
constexpr size_t kFirstTabBytes = 2 * (kFirstTabDeg + 1) * kFirstTabDeg * sizeof(double);

// LDS carve-up (bytes), shared by host and device
__host__ __device__ inline size_t lds_bytes_needed(int s, int n, int nnz, int S, bool want_llr)
{
    size_t b = (size_t)nnz * S * 8;             // messages
    if (want_llr) b += (size_t)n * S * 8;       // LLRs of the active syndromes
    b += (size_t)s * 8 + (size_t)n * 8;         // syndrome masks, decision masks
    b += 64 * 8;                                // per-wave reduction words + control
    b += 2 * ((size_t)s + 1 + (size_t)n + 1 + 2 * (size_t)nnz) + 16;   // the Tanner graph itself (uint16:
                                                                        // nnz*8 B must fit the LDS, so every index < 65536)
    b += kFirstTabBytes + 8;                    // check -> bit messages of iteration 1 by (sign, degree, position)
    return b;
}

template <int DC>
__device__ __forceinline__ void lds_check_unit(double *M, int S, int deg, double sigma, bool first, double r)
{
    // M points at msg[(e0)*S + sigma_lane]; rows are S doubles apart
    if (deg <= DC) {
        double a[DC], pre[DC];
#pragma unroll
        for (int k = 0; k < DC; ++k)
            if (k < deg) {
                const double m = first ? r : M[(size_t)k * S];
                a[k] = 2.0 / (1.0 + m) - 1.0;                         // :140 / :148
            }
        double P = sigma;                                             // :136
#pragma unroll
        for (int k = 0; k < DC; ++k)
            if (k < deg) { pre[k] = P; P = P * a[k]; }                // :139-140
        double Sx = 1.0;                                              // :143
#pragma unroll
        for (int k = DC - 1; k >= 0; --k)
            if (k < deg) {
                const double t = pre[k] * Sx;                         // :146
                M[(size_t)k * S] = (1.0 - t) / (1.0 + t);             // :147
                Sx = Sx * a[k];                                       // :148
            }
    } else {
        double Sx = 1.0;
        for (int k = deg - 1; k >= 0; --k) {
            double P = sigma;
            for (int q = 0; q < k; ++q) {
                const double m = first ? r : M[(size_t)q * S];
                P = P * (2.0 / (1.0 + m) - 1.0);
            }
            const double mk = first ? r : M[(size_t)k * S];
            const double ak = 2.0 / (1.0 + mk) - 1.0;
            const double t = P * Sx;
            M[(size_t)k * S] = (1.0 - t) / (1.0 + t);
            Sx = Sx * ak;
        }
    }
}

template <int DV, typename IDX>
__device__ __forceinline__ double lds_bit_unit(double *Ms, int S, const IDX *pos, int deg, double r)
{
    // Ms points at msg[sigma_lane]; edge e lives at Ms[e*S]
    double F = r;                                                     // :153
    if (deg <= DV) {
        double c[DV], pre[DV];
        int at[DV];
#pragma unroll
        for (int k = 0; k < DV; ++k)
            if (k < deg) { at[k] = (int)pos[k] * S; c[k] = Ms[at[k]]; }
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
                Ms[at[k]] = pre[k] * G;                               // :172
                G = G * c[k];                                         // :173
                if (G != G) G = 1.0;                                  // :174-176
            }
    } else {
        for (int k = 0; k < deg; ++k) {
            F = F * Ms[(int)pos[k] * S];
            if (F != F) F = 1.0;
        }
        double G = 1.0;
        for (int k = deg - 1; k >= 0; --k) {
            double Pk = r;
            for (int q = 0; q < k; ++q) {
                Pk = Pk * Ms[(int)pos[q] * S];
                if (Pk != Pk) Pk = 1.0;
            }
            const int a = (int)pos[k] * S;
            const double ck = Ms[a];
            Ms[a] = Pk * G;
            G = G * ck;
            if (G != G) G = 1.0;
        }
    }
    return F;
}

template <int DC, int DV, bool WANT_LLR, int THREADS>
__global__ void __launch_bounds__(THREADS)
bp_lds_kernel(LdsParams p, const int *__restrict__ g_row_ptr, const int *__restrict__ g_edge_bit,
              const int *__restrict__ g_col_ptr, const int *__restrict__ g_csc2csr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int s = p.s, n = p.n, nnz = p.nnz;
    const int logS = p.logS, S = 1 << logS;
    double *M = (double *)lds_raw;                              // [nnz][S]
    double *L = M + (size_t)nnz * S;                            // [n][S]   (WANT_LLR only)
    u64 *sbits = (u64 *)(L + (WANT_LLR ? (size_t)n * S : 0));   // [s]  bit sigma = syndrome entry parity
    u64 *ebits = sbits + s;                                     // [n]  bit sigma = hard decision
    u64 *red = ebits + n;                                       // [THREADS/64] + control words
    typedef unsigned short idx_t;
    idx_t *row_ptr = (idx_t *)(red + 64);                       // [s+1]  the graph, LDS copies
    idx_t *edge_bit = row_ptr + (s + 1);                        // [nnz]
    idx_t *col_ptr = edge_bit + nnz;                            // [n+1]
    idx_t *csc2csr = col_ptr + (n + 1);                         // [nnz]
    double *ftab = (double *)(((uintptr_t)(csc2csr + nnz) + 7) & ~(uintptr_t)7);   // [2][kFirstTabDeg + 1][kFirstTabDeg]
    __shared__ int sh_group;
    for (int t = threadIdx.x; t < 2 * (kFirstTabDeg + 1); t += THREADS) {   // (sign, degree) -> the messages of iteration 1
        const int neg = t & 1, d = t >> 1;
        if (d <= DC) lds_check_unit<DC>(ftab + (size_t)(neg * (kFirstTabDeg + 1) + d) * kFirstTabDeg, 1, d, neg ? -1.0 : 1.0, true, p.r);
    }
    for (int i = threadIdx.x; i <= s; i += THREADS) row_ptr[i] = (idx_t)g_row_ptr[i];
    for (int i = threadIdx.x; i <= n; i += THREADS) col_ptr[i] = (idx_t)g_col_ptr[i];
    for (int i = threadIdx.x; i < nnz; i += THREADS) { edge_bit[i] = (idx_t)g_edge_bit[i]; csc2csr[i] = (idx_t)g_csc2csr[i]; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int W = THREADS / 64;
    const double r = p.r;
    const u64 maskS = (S == 64) ? ~0ull : ((1ull << S) - 1ull);
    const int sig = tid & (S - 1);                              // this lane's syndrome slot (THREADS % S == 0)

    u64 acc_iters = 0, acc_check = 0, acc_var = 0, acc_rest = 0;   // per workgroup, flushed once at exit
    int g_next = 0, g_end = 0;                                    // the chunk of groups this workgroup holds
    for (;;) {
        if (g_next >= g_end) {
            if (!p.queue) {                   // latency mode: exactly one group per workgroup
                if (g_end != 0 || (int)blockIdx.x >= p.ngroups) break;
                g_next = (int)blockIdx.x;
                g_end = g_next + 1;
                __syncthreads();              // the graph copy above
            } else {
                if (tid == 0) sh_group = (int)atomicAdd(p.queue, (unsigned)p.chunk);
                __syncthreads();
                g_next = sh_group;
                g_end = min(g_next + p.chunk, p.ngroups);
                __syncthreads();
                if (g_next >= p.ngroups) break;   // every wave of every workgroup reaches this
            }
        }
        const int g = g_next++;
        const u64 tg0 = LDS_CLOCK();
        u64 tk_check = 0, tk_var = 0;
        const long long b0 = (long long)g << logS;
        const long long left = p.batch - b0;
        const u64 valid = left >= S ? maskS : ((1ull << left) - 1ull);

        // ---- syndromes in: one thread per check, S byte reads each (coalesced along the check index)
        u64 never_l = 0;
        for (int i = tid; i < s; i += THREADS) {
            u64 m = 0;
            for (int q = 0; q < S; ++q)
                if ((valid >> q) & 1ull) {
                    const unsigned v = p.syn[(size_t)(b0 + q) * s + i];
                    m |= (u64)(v & 1u) << q;
                    never_l |= (u64)(v > 1u) << q;
                }
            sbits[i] = m;
        }
        never_l = wave_or(never_l);
        if (lane == 0) red[w] = never_l;
        __syncthreads();
        u64 never = 0;
#pragma unroll
        for (int q = 0; q < W; ++q) never |= red[q];
        __syncthreads();

        u64 active = valid;
        u64 conv_mask = 0;
        int my_iters = 0;  // meaningful in threads tid < S
        int it = 0;
        while (active != 0 && it < p.max_iters) {
            ++it;
            const bool first = (it == 1);
            const u64 t0 = LDS_CLOCK();
            // ---- check sweep: unit u = (check u >> logS, syndrome u & (S-1))
            for (int u = tid; u < (s << logS); u += THREADS) {
                const int i = u >> logS;
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const bool neg = (sbits[i] >> sig) & 1ull;                       // (-1)^syndrome[i]  :136
                if (first && deg <= kFirstTabDeg && deg <= DC) {
                    const double *tr = ftab + (size_t)((neg ? 1 : 0) * (kFirstTabDeg + 1) + deg) * kFirstTabDeg;
                    double *Mi = M + (size_t)e0 * S + sig;
                    for (int k = 0; k < deg; ++k) Mi[(size_t)k * S] = tr[k];
                } else {
                    lds_check_unit<DC>(M + (size_t)e0 * S + sig, S, deg, neg ? -1.0 : 1.0, first, r);
                }
            }
            __syncthreads();
            const u64 t1 = LDS_CLOCK();
            // ---- variable sweep
            for (int u0 = w * 64; u0 < (n << logS); u0 += THREADS) {
                const int u = u0 + lane;
                const bool in = u < (n << logS);
                const int j = in ? (u >> logS) : 0;
                double T = 0.0;
                if (in) {
                    const int c0 = col_ptr[j];
                    const int deg = col_ptr[j + 1] - c0;
                    T = lds_bit_unit<DV, idx_t>(M + sig, S, csc2csr + c0, deg, r);
                    if (WANT_LLR) {
                        if ((active >> sig) & 1ull) L[(size_t)j * S + sig] = llr_of(T, p.llr_exact);   // :163
                    }
                }
                const u64 bal = __ballot(in && (T >= 1.0));                       // :164-168
                if (in && sig == 0) {
                    const u64 dec = (bal >> (lane & ~(S - 1))) & maskS;
                    u64 v = dec;
                    if (!first) v = (ebits[j] & ~active) | (dec & active);        // frozen syndromes keep theirs
                    ebits[j] = v;
                }
            }
            __syncthreads();
            const u64 t2 = LDS_CLOCK();
            tk_check += t1 - t0;
            tk_var += t2 - t1;
            // ---- convergence test (:180-184): one thread per check
            u64 mism = 0;
            for (int i = tid; i < s; i += THREADS) {
                u64 par = 0;
                const int e1 = row_ptr[i + 1];
                for (int e = row_ptr[i]; e < e1; ++e) par ^= ebits[edge_bit[e]];
                mism |= par ^ sbits[i];
            }
            mism = wave_or(mism);
            if (lane == 0) red[w] = mism;
            __syncthreads();
            u64 U = never;
#pragma unroll
            for (int q = 0; q < W; ++q) U |= red[q];
            U &= maskS;
            const u64 newly = active & ~U;
            if (tid < S && ((newly >> tid) & 1ull)) my_iters = it;
            conv_mask |= newly;
            active &= U;
            __syncthreads();   // red[] is rewritten next iteration
        }
        if (tid < S && ((active >> tid) & 1ull)) my_iters = it;

        // ---- results out (coalesced along the bit index)
        for (int idx = tid; idx < (n << logS); idx += THREADS) {
            const int q = idx / n, j = idx - q * n;
            if ((valid >> q) & 1ull) {
                p.err[(size_t)(b0 + q) * n + j] = (unsigned char)((ebits[j] >> q) & 1ull);
                if (WANT_LLR) p.llr[(size_t)(b0 + q) * n + j] = L[(size_t)j * S + q];
            }
        }
        if (tid < S && ((valid >> tid) & 1ull)) {
            p.conv[b0 + tid] = (unsigned char)((conv_mask >> tid) & 1ull);
            if (p.iters) p.iters[b0 + tid] = my_iters;
        }
        if (w == 0) {
            int tot = (tid < S && ((valid >> tid) & 1ull)) ? my_iters : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
            if (lane == 0) {
                const u64 tg1 = LDS_CLOCK();
                acc_iters += (u64)tot;
                acc_check += tk_check;
                acc_var += tk_var;
                acc_rest += (tg1 - tg0) - tk_check - tk_var;
            }
        }
        __syncthreads();
    }
    if (p.done_flag) publish_done(p.done_count, p.done_flag, p.done_ticket);
    if (p.next_ctrl && blockIdx.x == 0 && tid < 8) p.next_ctrl[tid] = 0;   // (at the end: up front it cost 6 VGPRs and a workgroup per CU)
    if (tid == 0 && p.sum_iters) {
        atomicAdd(p.sum_iters, acc_iters);
        if (LDPC_LDS_STAMPS) {
            atomicAdd(&p.phase_ticks[0], acc_check);
            atomicAdd(&p.phase_ticks[1], acc_var);
            atomicAdd(&p.phase_ticks[2], acc_rest);
        }
    }
}

}  // namespace ldpc
