// bpots_kernels.hpp -- BP-OTS (belief propagation with oscillation-driven biasing,
// src/decoders/bpots_decoder.jl) on gfx950, LDS-resident like bp_lds_kernels.hpp:
// a workgroup keeps every message, LLR and oscillation counter of S syndromes in LDS and runs
// all iterations with workgroup barriers only; a work item is one (node, syndrome) pair.
//
// Reference statements reproduced, order of every sum / product included:
//   reset! + priors            :142-154, :231-237
//   update_variable_to_check!  :158-172   left fold over the OTHER checks, then Omega + sum
//   update_check_to_variable!  :178-210   clamp(tanh(v/2)), left fold over the OTHER bits, sign,
//                                         clamp, 2 atanh, clamp to +-100
//   compute_beliefs!           :120-136   Omega + left fold, decision = llr < 0
//   oscillation / best-solution bookkeeping and the bias step  :257-338
// tanh / atanh are portable_math.h's (shared with the CPU oracle, so data-dependent decisions --
// sign of an LLR, arg-min |LLR| -- are identical on CPU and GPU; see that header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "portable_math.h"
#include "latency_mode.hpp"

namespace ldpc {

typedef unsigned long long u64;

struct OtsParams {
    int s, n, nnz;
    int max_iters, T;
    int logS, ngroups, chunk;
    long long batch;
    double prior;   // log((1 - 2p/3) / (2p/3)), computed on the host (:231)
    double C;
    const unsigned char *syn;   // [batch][s]
    unsigned char *err;         // [batch][n]  best_decisions
    unsigned char *conv;        // [batch]
    int *iters;                 // [batch] or nullptr
    unsigned int *queue;        // nullptr = latency mode: workgroup g decodes group g only (gridDim.x == ngroups)
    unsigned int *done_count;   // latency mode: device word, zero between launches
    unsigned int *done_flag;    // host-mapped word the last workgroup stores done_ticket into (nullptr = nobody waits)
    unsigned int done_ticket;
};

constexpr int kOtsThreads = 512;

__host__ __device__ inline size_t ots_lds_bytes(int s, int n, int nnz, int S)
{
    size_t b = (size_t)nnz * S * 8;                 // messages (in place: cv <-> vc)
    b += (size_t)n * S * 8;                         // llrs
    b += (size_t)n * S * 4;                         // oscillation counters
    b += 3 * (size_t)n * 8;                         // decision / prior / best masks
    b += 4 * (size_t)s * 8;                         // sign, target, never, parity masks
    b += 8 * 64 * 4 + 64;                           // per-syndrome words
    b += (size_t)kOtsThreads * 24;                  // arg-min partials (one double key + three ints per thread)
    b += 2 * ((size_t)s + 1 + (size_t)n + 1 + 2 * (size_t)nnz) + 32;   // graph (uint16)
    return b;
}

template <int DC, int DV>
__global__ void __launch_bounds__(kOtsThreads)
bpots_lds_kernel(OtsParams p, const int *__restrict__ g_row_ptr, const int *__restrict__ g_csc_row,
                 const int *__restrict__ g_col_ptr, const int *__restrict__ g_csc2csr)
{
    constexpr int THREADS = kOtsThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int s = p.s, n = p.n, nnz = p.nnz;
    const int logS = p.logS, S = 1 << logS;
    double *M = (double *)lds_raw;                        // [nnz][S]
    double *LLR = M + (size_t)nnz * S;                    // [n][S]
    double *pkey = LLR + (size_t)n * S;                   // [THREADS] arg-min partial keys
    u64 *dec = (u64 *)(pkey + THREADS);                   // [n]
    u64 *prior = dec + n;                                 // [n]
    u64 *best = prior + n;                                // [n]
    u64 *sgn = best + n;                                  // [s] syndrome entry != 0  (:195)
    u64 *tgt = sgn + s;                                   // [s] syndrome entry == 1
    u64 *nev = tgt + s;                                   // [s] syndrome entry > 1: can never match (:273)
    u64 *par = nev + s;                                   // [s]
    u64 *words = par + s;                                 // [8]: 0 upd, 1 conv, 2 bias
    int *OSC = (int *)(words + 8);                        // [n][S]
    int *posc = OSC + (size_t)n * S;                      // [THREADS] partial osc
    int *pidx = posc + THREADS;                           // [THREADS] partial idx (j1 scan)
    int *pidx2 = pidx + THREADS;                          // [THREADS] partial idx (j2 scan)
    int *cnt_m = pidx2 + THREADS;                         // [64]
    int *cnt_w = cnt_m + 64;                              // [64]
    int *best_m = cnt_w + 64;                             // [64]
    int *best_w = best_m + 64;                            // [64]
    int *bj1 = best_w + 64;                               // [64] biased node 1 (-1 none)
    int *bj2 = bj1 + 64;                                  // [64]
    typedef unsigned short idx_t;
    idx_t *row_ptr = (idx_t *)(bj2 + 64);                 // [s+1]
    idx_t *csc_row = row_ptr + (s + 1);                   // [nnz]
    idx_t *col_ptr = csc_row + nnz;                       // [n+1]
    idx_t *csc2csr = col_ptr + (n + 1);                   // [nnz]
    __shared__ int sh_group;
    __shared__ double pkeyB[THREADS];                     // second key array (|llr| for the j2 scan)

    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i <= s; i += THREADS) row_ptr[i] = (idx_t)g_row_ptr[i];
    for (int i = tid; i <= n; i += THREADS) col_ptr[i] = (idx_t)g_col_ptr[i];
    for (int i = tid; i < nnz; i += THREADS) { csc_row[i] = (idx_t)g_csc_row[i]; csc2csr[i] = (idx_t)g_csc2csr[i]; }
    const u64 maskS = (S == 64) ? ~0ull : ((1ull << S) - 1ull);
    const int sig = tid & (S - 1);
    const double PI0 = p.prior, NEGC = -p.C;
    const double MAX_TANH = 0.99999, MAX_MSG = 100.0;

    int g_next = 0, g_end = 0;
    for (;;) {
        if (g_next >= g_end) {
            if (!p.queue) {
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
                if (g_next >= p.ngroups) break;
            }
        }
        const int g = g_next++;
        const long long b0 = (long long)g << logS;
        const long long left = p.batch - b0;
        const u64 valid = left >= S ? maskS : ((1ull << left) - 1ull);

        // ---- reset! (:142-154) and syndromes in
        for (int i = tid; i < s; i += THREADS) {
            u64 a = 0, b = 0, c = 0;
            for (int q = 0; q < S; ++q)
                if ((valid >> q) & 1ull) {
                    const unsigned v = p.syn[(size_t)(b0 + q) * s + i];
                    a |= (u64)(v != 0u) << q;
                    b |= (u64)(v == 1u) << q;
                    c |= (u64)(v > 1u) << q;
                }
            sgn[i] = a; tgt[i] = b; nev[i] = c;
        }
        for (int e = tid; e < (nnz << logS); e += THREADS) M[e] = 0.0;
        for (int e = tid; e < (n << logS); e += THREADS) OSC[e] = 0;
        for (int j = tid; j < n; j += THREADS) { prior[j] = 0; best[j] = 0; }
        if (tid < 64) { best_m[tid] = s; best_w[tid] = n; bj1[tid] = -1; bj2[tid] = -1; }   // :236-237
        __syncthreads();

        u64 active = valid, conv_mask = 0;
        int my_iters = 0;
        int it = 0;
        while (active != 0 && it < p.max_iters) {
            ++it;
            // ---- variable -> check (:241-245, :158-172)
            for (int u = tid; u < (n << logS); u += THREADS) {
                const int j = u >> logS;
                const int c0 = col_ptr[j];
                const int deg = (int)col_ptr[j + 1] - c0;
                const double om = (j == bj1[sig] || j == bj2[sig]) ? NEGC : PI0;
                double c[DV];
                int at[DV];
#pragma unroll
                for (int k = 0; k < DV; ++k)
                    if (k < deg) { at[k] = (int)csc2csr[c0 + k] * S + sig; c[k] = M[at[k]]; }
#pragma unroll
                for (int k = 0; k < DV; ++k)
                    if (k < deg) {
                        double sum = 0.0;
#pragma unroll
                        for (int q = 0; q < DV; ++q)
                            if (q < deg && q != k) sum += c[q];
                        M[at[k]] = om + sum;
                    }
            }
            if (tid < 64) { cnt_m[tid] = 0; cnt_w[tid] = 0; }
            if (tid < 8) words[tid] = 0;
            __syncthreads();
            // ---- check -> variable (:247-251, :178-210)
            for (int u = tid; u < (s << logS); u += THREADS) {
                const int i = u >> logS;
                const int e0 = row_ptr[i];
                const int deg = (int)row_ptr[i + 1] - e0;
                const bool flip = (sgn[i] >> sig) & 1ull;
                if (sig == 0) par[i] = 0;
                double t[DC];
                double *Mi = M + (size_t)e0 * S + sig;
#pragma unroll
                for (int k = 0; k < DC; ++k)
                    if (k < deg) {
                        double tv = pm_tanh(0.5 * Mi[(size_t)k * S]);
                        tv = tv > MAX_TANH ? MAX_TANH : (tv < -MAX_TANH ? -MAX_TANH : tv);   // min(MAX, max(-MAX, t))
                        t[k] = tv;
                    }
#pragma unroll
                for (int k = 0; k < DC; ++k)
                    if (k < deg) {
                        double prod = 1.0;
#pragma unroll
                        for (int q = 0; q < DC; ++q)
                            if (q < deg && q != k) prod *= t[q];
                        if (flip) prod = -prod;
                        if (pm_fabs(prod) >= MAX_TANH) prod = prod > 0 ? MAX_TANH : -MAX_TANH;
                        double msg = 2.0 * pm_atanh(prod);
                        msg = msg > MAX_MSG ? MAX_MSG : (msg < -MAX_MSG ? -MAX_MSG : msg);
                        Mi[(size_t)k * S] = msg;
                    }
            }
            __syncthreads();
            // ---- beliefs, decisions, oscillations, parities (:120-136, :257-263)
            for (int u0 = (tid & ~63); u0 < (n << logS); u0 += THREADS) {
                const int u = u0 + lane;
                const bool in = u < (n << logS);
                const int j = in ? (u >> logS) : 0;
                double llr = 0.0;
                int c0 = 0, deg = 0;
                if (in) {
                    c0 = col_ptr[j];
                    deg = (int)col_ptr[j + 1] - c0;
                    llr = (j == bj1[sig] || j == bj2[sig]) ? NEGC : PI0;
                    for (int k = 0; k < deg; ++k) llr += M[(int)csc2csr[c0 + k] * S + sig];
                    LLR[(size_t)j * S + sig] = llr;
                }
                const bool d1 = in && (llr < 0.0);
                const u64 bal = __ballot(d1);
                if (in) {
                    if (it > 1 && (d1 != (bool)((prior[j] >> sig) & 1ull))) OSC[(size_t)j * S + sig] += 1;
                }
                // every lane of this bit has read prior[j] above (same instruction stream) before lane sig==0 rewrites it
                if (in && sig == 0) {
                    const u64 dm = (bal >> (lane & ~(S - 1))) & maskS;
                    dec[j] = dm;
                    prior[j] = dm;
                    if (dm)
                        for (int k = 0; k < deg; ++k)
                            atomicXor((unsigned long long *)&par[csc_row[c0 + k]], (unsigned long long)dm);
                }
            }
            __syncthreads();
            // ---- mismatch and weight per syndrome (:266-281)
            for (int i = tid; i < s; i += THREADS) {
                u64 mis = ((par[i] ^ tgt[i]) | nev[i]) & maskS;
                while (mis) { const int q = __ffsll((long long)mis) - 1; mis &= mis - 1; atomicAdd(&cnt_m[q], 1); }
            }
            for (int j = tid; j < n; j += THREADS) {
                u64 dm = dec[j];
                while (dm) { const int q = __ffsll((long long)dm) - 1; dm &= dm - 1; atomicAdd(&cnt_w[q], 1); }
            }
            __syncthreads();
            // ---- best solution so far (:284-292), who converged, who gets biased (:295)
            if (tid < S && ((active >> tid) & 1ull)) {
                const int m = cnt_m[tid], wgt = cnt_w[tid];
                if (m < best_m[tid] || (m == best_m[tid] && wgt < best_w[tid])) {
                    best_m[tid] = m; best_w[tid] = wgt;
                    atomicOr((unsigned long long *)&words[0], 1ull << tid);
                    if (m == 0) atomicOr((unsigned long long *)&words[1], 1ull << tid);
                }
                if (m > 0 && (it % p.T) == 0) atomicOr((unsigned long long *)&words[2], 1ull << tid);
            }
            __syncthreads();
            const u64 updw = words[0], convw = words[1], biasw = words[2];
            for (int j = tid; j < n; j += THREADS) best[j] = (best[j] & ~updw) | (dec[j] & updw);
            if (tid < S && ((convw >> tid) & 1ull)) my_iters = it;
            conv_mask |= convw;
            active &= ~convw;
            if (biasw != 0) {   // wave-uniform: words[] is the same for everybody
                // ---- bias step (:297-337): Omega .= Pi, then j1 (most oscillating, then least |llr|, then
                //      lowest index) and j2 (least |llr| overall, lowest index)
                {
                    const int jstep = THREADS >> logS;
                    int bo = -1, bi = -1, bi2 = -1;
                    double bk = 0.0, bk2 = 0.0;
                    for (int j = tid >> logS; j < n; j += jstep) {
                        const int o = OSC[(size_t)j * S + sig];
                        const double a = pm_fabs(LLR[(size_t)j * S + sig]);
                        if (bi < 0 || o > bo || (o == bo && a < bk)) { bo = o; bk = a; bi = j; }
                        if (bi2 < 0 || a < bk2) { bk2 = a; bi2 = j; }
                    }
                    posc[tid] = bo; pkey[tid] = bk; pidx[tid] = bi; pkeyB[tid] = bk2; pidx2[tid] = bi2;
                }
                __syncthreads();
                if (tid < S && ((biasw >> tid) & 1ull)) {
                    int bo = -1, bi = -1, bi2 = -1;
                    double bk = 0.0, bk2 = 0.0;
                    for (int t = tid; t < THREADS; t += S) {     // partials of this syndrome slot
                        const int i1 = pidx[t];
                        if (i1 >= 0) {
                            const int o = posc[t];
                            const double a = pkey[t];
                            if (bi < 0 || o > bo || (o == bo && (a < bk || (a == bk && i1 < bi)))) { bo = o; bk = a; bi = i1; }
                        }
                        const int i2 = pidx2[t];
                        if (i2 >= 0) {
                            const double a2 = pkeyB[t];
                            if (bi2 < 0 || a2 < bk2 || (a2 == bk2 && i2 < bi2)) { bk2 = a2; bi2 = i2; }
                        }
                    }
                    bj1[tid] = -1; bj2[tid] = -1;                // Omega .= Pi (:297)
                    if (bo > 0) {                                // maximum(oscillations) > 0 (:300)
                        OSC[(size_t)bi * S + tid] = 0;           // :320
                        bj1[tid] = bi;                           // :323
                        bj2[tid] = bi2;                          // :336
                    }
                }
            }
            __syncthreads();
        }
        if (tid < S && ((active >> tid) & 1ull)) my_iters = it;

        // ---- results out: best_decisions (:340 / :291)
        for (int idx = tid; idx < (n << logS); idx += THREADS) {
            const int q = idx / n, j = idx - q * n;
            if ((valid >> q) & 1ull) p.err[(size_t)(b0 + q) * n + j] = (unsigned char)((best[j] >> q) & 1ull);
        }
        if (tid < S && ((valid >> tid) & 1ull)) {
            p.conv[b0 + tid] = (unsigned char)((conv_mask >> tid) & 1ull);
            if (p.iters) p.iters[b0 + tid] = my_iters;
        }
        __syncthreads();
    }
    if (p.done_flag) publish_done(p.done_count, p.done_flag, p.done_ticket);
}

// ---------------------------------------------------------------------------------------------
// BP-OTS for graphs beyond the LDS: one workgroup per syndrome, one thread per Tanner-graph node (the
// bp_node_kernels.hpp geometry).  The messages (nnz doubles, CSR order, updated in place), the LLRs and the
// oscillation counters of the syndrome live in a private slot of a global workspace, which the workgroup's CU keeps
// in its L2; syndrome codes, decisions and parities are bytes / words in LDS; the graph is read from global memory.
// Statement for statement the arithmetic of bpots_lds_kernel with S = 1 (same left folds, same clamps, same
// tie-breaks), so the two kernels -- and the oracle -- agree bit for bit.
// ---------------------------------------------------------------------------------------------
struct OtsNodeParams {
    int s, n, nnz;
    int max_iters, T;
    long long batch;
    double prior, C;
    const unsigned char *syn;
    unsigned char *err, *conv;
    int *iters;
    unsigned int *queue;
    double *ws;                 // [gridDim.x][slot_doubles]: M[nnz] | LLR[n] | OSC[n] (ints, in the tail)
    long long slot_doubles;
};

constexpr int kOtsNodeThreads = 1024;

__host__ __device__ inline size_t ots_node_lds_bytes(int s, int n)
{
    return (((size_t)s + 3 * (size_t)n + 15) & ~(size_t)15) + 4 * (size_t)s + (size_t)kOtsNodeThreads * 28 + 64;
}
__host__ __device__ inline size_t ots_node_slot_doubles(int n, int nnz)
{
    return (((size_t)nnz + (size_t)n + ((size_t)n + 1) / 2) + 63) & ~(size_t)63;
}

template <int DC, int DV>
__global__ void __launch_bounds__(kOtsNodeThreads)
bpots_node_kernel(OtsNodeParams p, const int *__restrict__ row_ptr, const int *__restrict__ csc_row,
                  const int *__restrict__ col_ptr, const int *__restrict__ csc2csr)
{
    constexpr int THREADS = kOtsNodeThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char ots_lds[];
    const int s = p.s, n = p.n, nnz = p.nnz;
    unsigned char *code = ots_lds;                          // [s] syndrome entry: 0, 1, or 2 (anything else)
    unsigned char *dec = code + s;                          // [n]
    unsigned char *prv = dec + n;                           // [n] decisions of the previous iteration
    unsigned char *best = prv + n;                          // [n]
    unsigned int *par = (unsigned int *)(ots_lds + (((size_t)s + 3 * (size_t)n + 15) & ~(size_t)15));   // [s]
    double *pk1 = (double *)(par + s + ((s & 1) ? 1 : 0));  // [THREADS] |llr| of this thread's j1 candidate
    double *pk2 = pk1 + THREADS;                            // [THREADS] |llr| of its j2 candidate
    int *po = (int *)(pk2 + THREADS);                       // [THREADS] oscillation count of the j1 candidate
    int *pi1 = po + THREADS;                                // [THREADS]
    int *pi2 = pi1 + THREADS;                               // [THREADS]
    __shared__ long long sh_b;
    __shared__ int sh_cnt_m, sh_cnt_w, sh_best_m, sh_best_w, sh_bj1, sh_bj2, sh_flags;   // flags: 1 update best, 2 converged, 4 bias
    double *M = p.ws + (size_t)blockIdx.x * (size_t)p.slot_doubles;
    double *LLR = M + nnz;
    int *OSC = (int *)(LLR + n);
    const int tid = threadIdx.x, lane = tid & 63;
    const double PI0 = p.prior, NEGC = -p.C;
    const double MAX_TANH = 0.99999, MAX_MSG = 100.0;

    for (;;) {
        if (tid == 0) {
            const long long q = (long long)atomicAdd(p.queue, 1u);
            sh_b = q >= p.batch ? -1 : q;
        }
        __syncthreads();
        const long long b = sh_b;
        if (b < 0) break;                                   // every wave of every workgroup reaches this
        // ---- reset! (:142-154) and the syndrome in
        for (int i = tid; i < s; i += THREADS) {
            const unsigned v = p.syn[(size_t)b * s + i];
            code[i] = (unsigned char)(v > 1u ? 2u : v);
        }
        for (int e = tid; e < nnz; e += THREADS) M[e] = 0.0;
        for (int j = tid; j < n; j += THREADS) { OSC[j] = 0; prv[j] = 0; best[j] = 0; }
        if (tid == 0) { sh_best_m = s; sh_best_w = n; sh_bj1 = -1; sh_bj2 = -1; }          // :236-237
        __syncthreads();

        int it = 0, converged = 0;
        while (it < p.max_iters) {
            ++it;
            const int bj1 = sh_bj1, bj2 = sh_bj2;
            // ---- variable -> check (:241-245, :158-172)
            for (int j = tid; j < n; j += THREADS) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                const double om = (j == bj1 || j == bj2) ? NEGC : PI0;
                double c[DV];
                int at[DV];
#pragma unroll
                for (int k = 0; k < DV; ++k)
                    if (k < deg) { at[k] = csc2csr[c0 + k]; c[k] = M[at[k]]; }
#pragma unroll
                for (int k = 0; k < DV; ++k)
                    if (k < deg) {
                        double sum = 0.0;
#pragma unroll
                        for (int q = 0; q < DV; ++q)
                            if (q < deg && q != k) sum += c[q];
                        M[at[k]] = om + sum;
                    }
            }
            if (tid == 0) { sh_cnt_m = 0; sh_cnt_w = 0; sh_flags = 0; }
            __syncthreads();
            // ---- check -> variable (:247-251, :178-210)
            for (int i = tid; i < s; i += THREADS) {
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const bool flip = code[i] != 0;
                par[i] = 0u;
                double t[DC];
                double *Mi = M + e0;
#pragma unroll
                for (int k = 0; k < DC; ++k)
                    if (k < deg) {
                        double tv = pm_tanh(0.5 * Mi[k]);
                        tv = tv > MAX_TANH ? MAX_TANH : (tv < -MAX_TANH ? -MAX_TANH : tv);
                        t[k] = tv;
                    }
#pragma unroll
                for (int k = 0; k < DC; ++k)
                    if (k < deg) {
                        double prod = 1.0;
#pragma unroll
                        for (int q = 0; q < DC; ++q)
                            if (q < deg && q != k) prod *= t[q];
                        if (flip) prod = -prod;
                        if (pm_fabs(prod) >= MAX_TANH) prod = prod > 0 ? MAX_TANH : -MAX_TANH;
                        double msg = 2.0 * pm_atanh(prod);
                        msg = msg > MAX_MSG ? MAX_MSG : (msg < -MAX_MSG ? -MAX_MSG : msg);
                        Mi[k] = msg;
                    }
            }
            __syncthreads();
            // ---- beliefs, decisions, oscillations, parities (:120-136, :257-263), weight (:281)
            int wcount = 0;
            for (int j = tid; j < n; j += THREADS) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                double llr = (j == bj1 || j == bj2) ? NEGC : PI0;
                for (int k = 0; k < deg; ++k) llr += M[csc2csr[c0 + k]];
                LLR[j] = llr;
                const unsigned char d1 = llr < 0.0 ? 1 : 0;
                if (it > 1 && d1 != prv[j]) OSC[j] += 1;
                dec[j] = d1;
                prv[j] = d1;
                if (d1) {
                    ++wcount;
                    for (int k = 0; k < deg; ++k) atomicXor(&par[csc_row[c0 + k]], 1u);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wcount += __shfl_xor(wcount, off, 64);
            if (lane == 0 && wcount) atomicAdd(&sh_cnt_w, wcount);
            __syncthreads();
            // ---- mismatch (:266-279)
            int mcount = 0;
            for (int i = tid; i < s; i += THREADS) mcount += (code[i] > 1 || (par[i] & 1u) != (unsigned)code[i]) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mcount += __shfl_xor(mcount, off, 64);
            if (lane == 0 && mcount) atomicAdd(&sh_cnt_m, mcount);
            __syncthreads();
            // ---- best solution so far (:284-292), converged?, bias due? (:295)
            if (tid == 0) {
                const int m = sh_cnt_m, wgt = sh_cnt_w;
                int f = 0;
                if (m < sh_best_m || (m == sh_best_m && wgt < sh_best_w)) {
                    sh_best_m = m; sh_best_w = wgt;
                    f |= 1;
                    if (m == 0) f |= 2;
                }
                if (m > 0 && (it % p.T) == 0) f |= 4;
                sh_flags = f;
            }
            __syncthreads();
            const int flags = sh_flags;
            if (flags & 1)
                for (int j = tid; j < n; j += THREADS) best[j] = dec[j];
            if (flags & 2) { converged = 1; __syncthreads(); break; }
            if (flags & 4) {
                // ---- bias step (:297-337): Omega .= Pi, then j1 (most oscillating, then least |llr|, then lowest
                //      index) and j2 (least |llr| overall, lowest index)
                int bo = -1, bi = -1, bi2 = -1;
                double bk = 0.0, bk2 = 0.0;
                for (int j = tid; j < n; j += THREADS) {
                    const int o = OSC[j];
                    const double a = pm_fabs(LLR[j]);
                    if (bi < 0 || o > bo || (o == bo && a < bk)) { bo = o; bk = a; bi = j; }
                    if (bi2 < 0 || a < bk2) { bk2 = a; bi2 = j; }
                }
                po[tid] = bo; pk1[tid] = bk; pi1[tid] = bi; pk2[tid] = bk2; pi2[tid] = bi2;
                __syncthreads();
                if (tid == 0) {
                    bo = -1; bi = -1; bi2 = -1; bk = 0.0; bk2 = 0.0;
                    for (int t = 0; t < THREADS; ++t) {
                        const int i1 = pi1[t];
                        if (i1 >= 0) {
                            const int o = po[t];
                            const double a = pk1[t];
                            if (bi < 0 || o > bo || (o == bo && (a < bk || (a == bk && i1 < bi)))) { bo = o; bk = a; bi = i1; }
                        }
                        const int i2 = pi2[t];
                        if (i2 >= 0) {
                            const double a2 = pk2[t];
                            if (bi2 < 0 || a2 < bk2 || (a2 == bk2 && i2 < bi2)) { bk2 = a2; bi2 = i2; }
                        }
                    }
                    sh_bj1 = -1; sh_bj2 = -1;                 // Omega .= Pi (:297)
                    if (bo > 0) {                             // maximum(oscillations) > 0 (:300)
                        OSC[bi] = 0;                          // :320
                        sh_bj1 = bi;                          // :323
                        sh_bj2 = bi2;                         // :336
                    }
                }
            }
            __syncthreads();
        }
        // ---- results out: best_decisions (:340 / :291)
        for (int j = tid; j < n; j += THREADS) p.err[(size_t)b * n + j] = best[j];
        if (tid == 0) {
            p.conv[b] = (unsigned char)converged;
            if (p.iters) p.iters[b] = it;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// BP-OTS without limits: graphs whose per-syndrome bytes do not fit a CU's LDS (n beyond ~30,000) or whose nodes are
// wider than the register buckets above (check degree > 32, bit degree > 16) -- the reference's BPOTSDecoder takes any
// H (bpots_decoder.jl:225-340).  The same geometry as bpots_node_kernel (one workgroup per syndrome, one thread per
// node), but EVERYTHING of the syndrome lives in its private global slot: messages M[nnz], a second array T[nnz] for
// what a node needs of its OLD values while it overwrites them (the "all other neighbours" folds of :161-167 and
// :182-192 are O(deg^2) loads from it instead of register arrays), LLRs, oscillation counters, parities, syndrome codes
// and the three decision vectors.  Only the per-thread candidates of the bias step are in LDS.  Statement for
// statement the arithmetic of the two kernels above -- the same left folds in the same order, the same clamps, the same
// tie-breaks --, so all three agree bit for bit wherever more than one applies (tests/test_gpu_bpots.py).
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t ots_big_slot_doubles(int s, int n, int nnz)
{
    // M[nnz] | T[nnz] | LLR[n] | OSC[n] (int) | par[s] (unsigned) | code[s] dec[n] prv[n] best[n] (bytes)
    const size_t bytes = (size_t)nnz * 16 + (size_t)n * 8 + (size_t)n * 4 + (size_t)s * 4 + (size_t)s + 3 * (size_t)n;
    return ((bytes + 7) / 8 + 63) & ~(size_t)63;
}

__global__ void __launch_bounds__(kOtsNodeThreads)
bpots_big_kernel(OtsNodeParams p, const int *__restrict__ row_ptr, const int *__restrict__ csc_row,
                 const int *__restrict__ col_ptr, const int *__restrict__ csc2csr)
{
    constexpr int THREADS = kOtsNodeThreads;
    __shared__ double pk1[THREADS], pk2[THREADS];
    __shared__ int po[THREADS], pi1[THREADS], pi2[THREADS];
    __shared__ long long sh_b;
    __shared__ int sh_cnt_m, sh_cnt_w, sh_best_m, sh_best_w, sh_bj1, sh_bj2, sh_flags;   // flags: 1 update best, 2 converged, 4 bias
    const int s = p.s, n = p.n, nnz = p.nnz;
    double *M = p.ws + (size_t)blockIdx.x * (size_t)p.slot_doubles;
    double *Told = M + nnz;
    double *LLR = Told + nnz;
    int *OSC = (int *)(LLR + n);
    unsigned int *par = (unsigned int *)(OSC + n);
    unsigned char *code = (unsigned char *)(par + s);
    unsigned char *dec = code + s, *prv = dec + n, *best = prv + n;
    const int tid = threadIdx.x, lane = tid & 63;
    const double PI0 = p.prior, NEGC = -p.C;
    const double MAX_TANH = 0.99999, MAX_MSG = 100.0;

    for (;;) {
        if (tid == 0) {
            const long long q = (long long)atomicAdd(p.queue, 1u);
            sh_b = q >= p.batch ? -1 : q;
        }
        __syncthreads();
        const long long b = sh_b;
        if (b < 0) break;                                   // every wave of every workgroup reaches this
        // ---- reset! (:142-154) and the syndrome in
        for (int i = tid; i < s; i += THREADS) {
            const unsigned v = p.syn[(size_t)b * s + i];
            code[i] = (unsigned char)(v > 1u ? 2u : v);
        }
        for (int e = tid; e < nnz; e += THREADS) M[e] = 0.0;
        for (int j = tid; j < n; j += THREADS) { OSC[j] = 0; prv[j] = 0; best[j] = 0; }
        if (tid == 0) { sh_best_m = s; sh_best_w = n; sh_bj1 = -1; sh_bj2 = -1; }          // :236-237
        __syncthreads();

        int it = 0, converged = 0;
        while (it < p.max_iters) {
            ++it;
            const int bj1 = sh_bj1, bj2 = sh_bj2;
            // ---- variable -> check (:241-245, :158-172): the old messages of the bit go to T, the new ones are folds over T
            for (int j = tid; j < n; j += THREADS) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                const double om = (j == bj1 || j == bj2) ? NEGC : PI0;
                for (int k = 0; k < deg; ++k) Told[c0 + k] = M[csc2csr[c0 + k]];
                for (int k = 0; k < deg; ++k) {
                    double sum = 0.0;
                    for (int q = 0; q < deg; ++q)
                        if (q != k) sum += Told[c0 + q];
                    M[csc2csr[c0 + k]] = om + sum;
                }
            }
            if (tid == 0) { sh_cnt_m = 0; sh_cnt_w = 0; sh_flags = 0; }
            __syncthreads();
            // ---- check -> variable (:247-251, :178-210): the clamped tanh of every edge goes to T first
            for (int i = tid; i < s; i += THREADS) {
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const bool flip = code[i] != 0;
                par[i] = 0u;
                for (int k = 0; k < deg; ++k) {
                    double tv = pm_tanh(0.5 * M[e0 + k]);
                    tv = tv > MAX_TANH ? MAX_TANH : (tv < -MAX_TANH ? -MAX_TANH : tv);
                    Told[e0 + k] = tv;
                }
                for (int k = 0; k < deg; ++k) {
                    double prod = 1.0;
                    for (int q = 0; q < deg; ++q)
                        if (q != k) prod *= Told[e0 + q];
                    if (flip) prod = -prod;
                    if (pm_fabs(prod) >= MAX_TANH) prod = prod > 0 ? MAX_TANH : -MAX_TANH;
                    double msg = 2.0 * pm_atanh(prod);
                    msg = msg > MAX_MSG ? MAX_MSG : (msg < -MAX_MSG ? -MAX_MSG : msg);
                    M[e0 + k] = msg;
                }
            }
            __syncthreads();
            // ---- beliefs, decisions, oscillations, parities (:120-136, :257-263), weight (:281)
            int wcount = 0;
            for (int j = tid; j < n; j += THREADS) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                double llr = (j == bj1 || j == bj2) ? NEGC : PI0;
                for (int k = 0; k < deg; ++k) llr += M[csc2csr[c0 + k]];
                LLR[j] = llr;
                const unsigned char d1 = llr < 0.0 ? 1 : 0;
                if (it > 1 && d1 != prv[j]) OSC[j] += 1;
                dec[j] = d1;
                prv[j] = d1;
                if (d1) {
                    ++wcount;
                    for (int k = 0; k < deg; ++k) atomicXor(&par[csc_row[c0 + k]], 1u);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wcount += __shfl_xor(wcount, off, 64);
            if (lane == 0 && wcount) atomicAdd(&sh_cnt_w, wcount);
            __syncthreads();
            // ---- mismatch (:266-279)
            int mcount = 0;
            for (int i = tid; i < s; i += THREADS) mcount += (code[i] > 1 || (par[i] & 1u) != (unsigned)code[i]) ? 1 : 0;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mcount += __shfl_xor(mcount, off, 64);
            if (lane == 0 && mcount) atomicAdd(&sh_cnt_m, mcount);
            __syncthreads();
            // ---- best solution so far (:284-292), converged?, bias due? (:295)
            if (tid == 0) {
                const int m = sh_cnt_m, wgt = sh_cnt_w;
                int f = 0;
                if (m < sh_best_m || (m == sh_best_m && wgt < sh_best_w)) {
                    sh_best_m = m; sh_best_w = wgt;
                    f |= 1;
                    if (m == 0) f |= 2;
                }
                if (m > 0 && (it % p.T) == 0) f |= 4;
                sh_flags = f;
            }
            __syncthreads();
            const int flags = sh_flags;
            if (flags & 1)
                for (int j = tid; j < n; j += THREADS) best[j] = dec[j];
            if (flags & 2) { converged = 1; __syncthreads(); break; }
            if (flags & 4) {
                // ---- bias step (:297-337): Omega .= Pi, then j1 (most oscillating, then least |llr|, then lowest
                //      index) and j2 (least |llr| overall, lowest index)
                int bo = -1, bi = -1, bi2 = -1;
                double bk = 0.0, bk2 = 0.0;
                for (int j = tid; j < n; j += THREADS) {
                    const int o = OSC[j];
                    const double a = pm_fabs(LLR[j]);
                    if (bi < 0 || o > bo || (o == bo && a < bk)) { bo = o; bk = a; bi = j; }
                    if (bi2 < 0 || a < bk2) { bk2 = a; bi2 = j; }
                }
                po[tid] = bo; pk1[tid] = bk; pi1[tid] = bi; pk2[tid] = bk2; pi2[tid] = bi2;
                __syncthreads();
                if (tid == 0) {
                    bo = -1; bi = -1; bi2 = -1; bk = 0.0; bk2 = 0.0;
                    for (int t = 0; t < THREADS; ++t) {
                        const int i1 = pi1[t];
                        if (i1 >= 0) {
                            const int o = po[t];
                            const double a = pk1[t];
                            if (bi < 0 || o > bo || (o == bo && (a < bk || (a == bk && i1 < bi)))) { bo = o; bk = a; bi = i1; }
                        }
                        const int i2 = pi2[t];
                        if (i2 >= 0) {
                            const double a2 = pk2[t];
                            if (bi2 < 0 || a2 < bk2 || (a2 == bk2 && i2 < bi2)) { bk2 = a2; bi2 = i2; }
                        }
                    }
                    sh_bj1 = -1; sh_bj2 = -1;                 // Omega .= Pi (:297)
                    if (bo > 0) {                             // maximum(oscillations) > 0 (:300)
                        OSC[bi] = 0;                          // :320
                        sh_bj1 = bi;                          // :323
                        sh_bj2 = bi2;                         // :336
                    }
                }
            }
            __syncthreads();
        }
        // ---- results out: best_decisions (:340 / :291)
        for (int j = tid; j < n; j += THREADS) p.err[(size_t)b * n + j] = best[j];
        if (tid == 0) {
            p.conv[b] = (unsigned char)converged;
            if (p.iters) p.iters[b] = it;
        }
        __syncthreads();
    }
}

}  // namespace ldpc
