// bp_kernels.hpp -- gfx950 (MI355X, CDNA4) device code for the belief-propagation
// message sweeps of LDPCDecoders.jl's BeliefPropagationDecoder.
//
// What is computed (reference: src/decoders/belief_propagation.jl):
//   check-node sweep    :135-150   ordered prefix x suffix product in the odds domain
//   variable-node sweep :152-178   ordered prefix x suffix product with NaN reset,
//                                  posterior odds, hard decision, log(1/T)
//   convergence test    :180-184   GF(2) parity of the hard decisions per check
//
// How it is laid out for the machine (nothing like the reference's dense s x n
// matrices):
//   * lane = syndrome.  A TILE is 64 syndromes = one wavefront wide.  Every
//     edge message of a tile is one 512-byte row  msg[edge][64]  so any edge
//     order is a perfectly coalesced wave access, and the serial, order-
//     dependent products of the reference run unchanged inside one lane --
//     bit-exactness by construction, no cross-lane re-association anywhere.
//   * ONE message array per tile, updated IN PLACE: a check owns its edges
//     during the check sweep (reads bit->check, writes check->bit into the same
//     rows), a bit owns its edges during the variable sweep.  Rows are stored
//     check-major (CSR order), so the check sweep streams contiguously and the
//     variable sweep gathers/scatters whole 512-byte rows.
//   * hard decisions and syndromes are 64-bit lane masks (one word per bit /
//     per check and tile): the convergence test is XORs of words.
//   * a workgroup owns a workspace slot and pulls tiles from a queue; it keeps
//     iterating on its tile with workgroup barriers only (syndromes are
//     independent, so there is no grid-wide synchronisation at all).
//
// Compiled with -ffp-contract=off: the reference (Julia) never fuses a*b+c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ldpc {

typedef unsigned long long u64;

constexpr int kTile = 64;  // syndromes per tile == wavefront width on gfx950

// Build-time tuning knobs (defaults = the shipped configuration; DESIGN.md lists what was measured)
#ifndef LDPC_MIN_WAVES   // 2nd __launch_bounds__ argument = waves per SIMD the register budget must allow.
#define LDPC_MIN_WAVES 6 // 6 => <=80 VGPRs => three 512-thread workgroups per CU (measured best: +9.6 % over 4)
#endif
#ifndef LDPC_STM_SC1     // 1 = message rows are stored write-through, agent-coherent (the team kernels: pick_team.hip)
#define LDPC_STM_SC1 0
#endif
#ifndef LDPC_NT          // 1 = non-temporal loads/stores for the streamed edge messages
#define LDPC_NT 0
#endif
#ifndef LDPC_ROTATE      // 1 = every workgroup starts its sweeps at a different node (de-phases the workgroups)
#define LDPC_ROTATE 1    // together with the pad: +1.3..1.4 % on C3 full-50 (two boxes, A/B in one process each)
#endif
#ifndef LDPC_PHASE_TICKS // 1 = the tile kernel stamps its phases (four 100 MHz clock reads per iteration, three atomics per
#define LDPC_PHASE_TICKS 1 // tile) for bench.py's phase shares; A/B against 0 on C3 full-50: see DESIGN.md "Measured"
#endif
#ifndef LDPC_SLOT_PAD    // bytes added to the workspace slot stride (breaks power-of-two slot strides)
#define LDPC_SLOT_PAD 1053184   // 1 MiB + 4.5 KiB
#endif

// Register budget per kernel variant, as the waves-per-SIMD the launch bound must admit.
// Narrow-degree variants (the regular LDPC codes of the benchmarks) run three 512-thread
// workgroups per CU; the wide register buckets need the registers more than the occupancy.
template <int DC, int DV, int THREADS>
constexpr int min_waves_per_simd()
{
    constexpr int one_block = THREADS / 256;  // waves per SIMD of a single workgroup
#ifndef LDPC_MIN_WAVES_MID
#define LDPC_MIN_WAVES_MID 3  // 16-wide bucket: 159 VGPRs, no scratch (4 would spill 96 B/lane; same speed, HBM-bound)
#endif
    int want = (DC <= 8 && DV <= 4) ? LDPC_MIN_WAVES : (DC <= 16 ? LDPC_MIN_WAVES_MID : 2);
    if (want < one_block) want = one_block;
    if (THREADS >= 1024 && want > 4) want = 4;
    return want;
}

// After the row loads of a node group (experiment, -DLDPC_LOADS_FIRST=1): keep the machine scheduler from sinking the
// later loads below the first node's arithmetic.  It does sink them -- the second check's 8 loads of a pair come ~130
// instructions after the first's -- and forcing all 16 to the front measured NOTHING on any workload
// (profiles/r03_loads_first_ab.txt): a wave's memory parallelism is not what bounds the team kernel's sweeps.
#ifndef LDPC_LOADS_FIRST
#define LDPC_LOADS_FIRST 0
#endif
__device__ __forceinline__ void loads_first()
{
#if LDPC_LOADS_FIRST
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ double ldm(const double *p)
{
#if LDPC_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void stm(double *p, double v)
{
#if LDPC_STM_SC1   // (write-through, agent-coherent: the team kernels, pick_team.hip)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif LDPC_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Parameters a pass touches once per tile (or once per hand-off).  They live in DEVICE memory (the host stores a
// block per pass with store_cold_kernel) and are read where they are needed: as by-value kernel arguments they
// were all loaded in the prologue and kept live through the sweeps, and the scalar registers that cost pushed the
// hot instantiation over its 80-VGPR budget (spilled SGPRs are parked in VGPR lanes).
struct BPCold {
    int *iters;                 // [batch] or nullptr
    unsigned char *conv;        // [batch]
    u64 *sum_iters;             // accumulated iterations executed
    u64 *phase_ticks;           // [3] 100 MHz ticks spent in check sweep / variable sweep / convergence test,
                                //     summed over workgroups (wave 0 of each; diagnostics for DESIGN.md)
    // Straggler hand-off (DESIGN.md "early exit"): a tile whose active lanes have dwindled to <= defer_thresh
    // gives those syndromes up WITH their message state: the lanes' columns of the tile's message rows are copied
    // into packed tiles of the next LEVEL (next_state, tile layout again: syndrome q of the level sits in lane
    // q % 64 of packed tile q / 64), their batch positions and iteration counts are appended to defer_list /
    // defer_it, and a later pass resumes them at the iteration they had reached -- nothing is decoded twice.
    int *defer_list;            // [next_cap] batch positions handed to the next level
    int *defer_it;              // [next_cap] iterations those syndromes have run
    unsigned int *defer_count;  // number of entries in defer_list (reserved by compare-and-swap: never beyond next_cap)
    double *next_state;         // [next_cap / 64][next_stride] packed message tiles of the next level
    long long next_stride;      // doubles between two packed tiles (>= nnz * 64)
    unsigned int next_cap;      // syndromes the next level can take; a tile that finds no room carries on by itself
    // A pass over a packed level (index != nullptr):
    const int *index;           // batch position of compact syndrome q
    const int *it0;             // iterations compact syndrome q has already run (its messages are in its packed tile)
};

struct BPParams {
    int s, n, nnz;
    int max_iters;
    int ntiles;
    long long batch;
    double r;  // channel odds per/(1-per)   (belief_propagation.jl:129,153)
    // per-launch buffers
    double *msg;          // [slots][slot_stride]   workspace (fresh tiles) or the packed tiles of a level
    long long slot_stride;  // doubles between consecutive slots (>= nnz*64)
    u64 *errmask;         // [ntiles][n]  hard decisions of the current iteration, every lane (bit l = syndrome 64*tile + l)
    u64 *finmask;         // [ntiles][n]  ... of the iteration at which each lane stopped (converged / out of iterations):
                          //              what the caller gets.  Captured from errmask when a lane stops.
    double *llr;          // [ntiles][n][64] or nullptr
    unsigned int *queue;  // tile queue head
    const BPCold *cold;   // everything touched once per tile (device memory)
    int defer_thresh;           // hand stragglers on once at most this many lanes are active; 0 = never (last level, or feature off)
    int defer_min_iter;         // do not give up before this many iterations of this pass
    const unsigned int *count_dev;  // a pass over a packed level: number of compact syndromes (device word; nullptr = p.batch)
    unsigned int count_skip;    // ... do nothing while *count_dev <= count_skip (bp_node_kernels.hpp takes those)
    int resumed;                // 1: a pass over a packed level (cold->index / cold->it0 are set, messages are in the packed tiles)
    int llr_exact;              // LLRs from the full posterior odds (llr_of)
};

// LLRs from 32 bits.  The reference's log_probabs[j] = log(1 / T) (:163) is wanted to 1e-5 (BASELINE.json north_star).
// The UPPER 32 BITS of the posterior odds T -- sign, exponent, the 20 leading fraction bits -- cost nothing to make (they
// are a register of T already) and halve what the team kernel has to write per bit and iteration (TeamParams::llr_raw =
// 4); the decoder puts the middle of what was cut off back (2^-21 relative: the LLR good to 5e-7; denormal odds with a finite
// LLR, 2^-1024 <= T < 2^-1022, LLR 708.4 ... 709.8, keep 18-19 significant bits: 2e-6), except under an all-ones exponent (+-infinity stays itself).
// T = +-0 comes back as a denormal whose reciprocal overflows, so log(1 / T) = +-Inf exactly as for 0; and every T
// below 2^-1024 -- where the reference's own 1 / T overflows to Inf -- still decodes below 2^-1024.
__device__ __forceinline__ unsigned int llr_hi32(double T) { return (unsigned int)__double2hiint(T); }
__device__ __forceinline__ double llr_from_hi32(unsigned int hi)
{
    return __hiloint2double((int)hi, (hi & 0x7FF00000u) == 0x7FF00000u ? 0 : (int)0x80000000u);
}

// log_probabs[j] as EVERY kernel of the library returns it (:163): log(1 / T) of the posterior odds T cut to their
// upper 32 bits (above) -- whatever kernel finishes a syndrome, its LLRs are the same bits, and within 5e-7 of the
// reference's, inside the 1e-5 of BASELINE.json -- unless the decoder was created with llr_exact (ldpc_bp_options):
// then log(1 / T) of T itself, as before round 4 (the team kernel then captures 8 bytes per bit and iteration, not 4).
// log(1 / v) for a v that llr_from_hi32 made (21 significant bits), to 6e-13 absolute -- a thousandth of a millionth of what
// the cut itself costs -- in some 40 instructions instead of the 150 of a division and the library's log: v = m 2^e with m
// in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s) for s = (m - 1) / (m + 1) (|s| < 0.172: seven terms), and the end cases as
// log(1 / v) has them: +Inf wherever the reference's own 1 / T overflows (v < 2^-1024: exact for these v, see above),
// -Inf for Inf, NaN for NaN and below zero.  Explicit fma()s: the same bits in every kernel.
__device__ __forceinline__ double llr_cut(double v)
{
    double m = __builtin_amdgcn_frexp_mant(v);            // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(v);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = fma(z, 1.0 / 13.0, 1.0 / 11.0);
    p = fma(z, p, 1.0 / 9.0);
    p = fma(z, p, 1.0 / 7.0);
    p = fma(z, p, 1.0 / 5.0);
    p = fma(z, p, 1.0 / 3.0);
    p = fma(z, p, 1.0);
    double l = -fma((double)e, 0.69314718055994530942, (s + s) * p);
    l = (v == __builtin_inf()) ? -__builtin_inf() : l;
    l = (v >= 0x1p-1024) ? l : (v >= 0.0 ? __builtin_inf() : __builtin_nan(""));
    return l;
}

__device__ __forceinline__ double llr_of(double T, int exact)
{
    return exact ? log(1.0 / T) : llr_cut(llr_from_hi32(llr_hi32(T)));
}

// a value known to be the same in every lane -> scalar registers
__device__ __forceinline__ u64 uniform64(u64 v)
{
    return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}

__device__ __forceinline__ u64 wave_or(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        v |= ((u64)hi << 32) | lo;
    }
    return v;
}

// ---------------------------------------------------------------------------
// The two fp64 divisions per edge of the check sweep (:140 = :148, and :147) are 22 of its ~41 vector instructions.
// hipcc expands an IEEE double division into v_div_scale x2, v_rcp, four FMAs (two Newton steps), a multiply, the
// residual FMA, v_div_fmas and v_div_fixup.  The two scale instructions only do something when an operand or the
// quotient is near the ends of the exponent range (zero, denormal, |d| >= 2^1022, exponent(n) - exponent(d) >= 768,
// a denormal quotient, |n| <= 2^-970), v_div_fmas is then a plain FMA, and v_div_fixup only replaces the result for
// zero / infinite / NaN operands -- for all other operands the quotient IS what the nine instructions in the middle
// compute, correctly rounded.  LDPC_FAST_DIV=1: where a whole wave's operands of a node are inside a range that
// rules all of that out (one compare per division, a wave-uniform branch per node) run those instructions alone
// (div_core), otherwise the full division: the same bits either way (tests/test_gpu_parity.py holds div_core
// against `/` on the range's edges, and every parity test runs through it).
//   :140  2 / (1 + m): d = 1 + m in [1, 2^500)  (m >= 0: odds)                -> quotient in (2^-499, 2]
//   :147  (1 - t) / (1 + t): |t| < 1, so both operands are in [2^-53, 2)       -> quotient in [2^-54, 2^54]
#ifndef LDPC_FAST_DIV
#define LDPC_FAST_DIV 0
#endif
__device__ __forceinline__ double div_core(double n, double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, y, q);
}
__device__ __forceinline__ bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }

// a[k] = 2 / (1 + m[k]) - 1  (:140 / :148: the same value both times)
template <int D>
__device__ __forceinline__ void check_factors(const double (&m)[D], double (&a)[D])
{
#if LDPC_FAST_DIV
    double d[D];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) { d[k] = 1.0 + m[k]; ok = ok && d[k] >= 1.0 && d[k] < 0x1p+500; }
    if (wave_all(ok)) {
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = div_core(2.0, d[k]) - 1.0;
        return;
    }
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = 2.0 / d[k] - 1.0;
#else
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = 2.0 / (1.0 + m[k]) - 1.0;
#endif
}
// out[k] = (1 - t[k]) / (1 + t[k])  (:147)
template <int D>
__device__ __forceinline__ void check_to_odds(const double (&t)[D], double (&out)[D])
{
#if LDPC_FAST_DIV
    bool ok = true;
#pragma unroll
    for (int k = 0; k < D; ++k) ok = ok && __builtin_fabs(t[k]) < 1.0;
    if (wave_all(ok)) {
#pragma unroll
        for (int k = 0; k < D; ++k) out[k] = div_core(1.0 - t[k], 1.0 + t[k]);
        return;
    }
#endif
#pragma unroll
    for (int k = 0; k < D; ++k) out[k] = (1.0 - t[k]) / (1.0 + t[k]);
}

// ---------------------------------------------------------------------------
// check-node update for one check and 64 syndromes (one wave).
// belief_propagation.jl:136-149.  `M` points at the check's first row + lane.
// D is the EXACT degree: straight-line code, all D row loads issued back to back.
// ---------------------------------------------------------------------------
// the arithmetic of a check once its D factors a[k] = 2/(1+m[k]) - 1 are known: ordered prefix and suffix products.
// TF ("t form"): store the product t itself and leave the division (1 - t) / (1 + t) of :147 to whoever loads the
// row next -- the variable sweep of the SAME kernel and iteration (bp_team_kernels.hpp: the check sweep of the
// persistent teams is bound by its 16 fp64 divisions per check, the variable sweep has arithmetic to spare).  The
// same operations on the same operands, so the same bits; what a sweep leaves behind at an iteration boundary (and
// hands to other kernels) is never in t form.
template <int D, bool TF = false>
__device__ __forceinline__ double check_finish_exact(double *M, const double (&a)[D], double sigma, double S0 = 1.0)
{
    double pre[D];
    double P = sigma;                                     // :136
#pragma unroll
    for (int k = 0; k < D; ++k) { pre[k] = P; P = P * a[k]; }          // :139-140
    double S = S0;                                        // :143 (1.0; check_update_halves: the product over the edges behind these)
#if LDPC_FAST_DIV
    double t[D], o[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) { t[k] = pre[k] * S; S = S * a[k]; }   // :146, :148
    if (TF) {
#pragma unroll
        for (int k = D - 1; k >= 0; --k) stm(M + (size_t)k * kTile, t[k]);
        return S;
    }
    check_to_odds<D>(t, o);                               // :147
#pragma unroll
    for (int k = D - 1; k >= 0; --k) stm(M + (size_t)k * kTile, o[k]);
#else
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        const double t = pre[k] * S;                      // :146
        stm(M + (size_t)k * kTile, TF ? t : (1.0 - t) / (1.0 + t));  // :147
        S = S * a[k];                                     // :148
    }
#endif
    return S;
}

// ... the same with the D new messages handed back instead of stored (rows that do not lie k * 64 doubles apart:
// bp_team_kernels.hpp keeps some in LDS)
template <int D, bool TF = false>
__device__ __forceinline__ void check_compute_exact(const double (&a)[D], double sigma, double (&out)[D])
{
    double pre[D];
    double P = sigma;                                     // :136
#pragma unroll
    for (int k = 0; k < D; ++k) { pre[k] = P; P = P * a[k]; }          // :139-140
    double S = 1.0;                                       // :143
#if LDPC_FAST_DIV
    double t[D];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) { t[k] = pre[k] * S; S = S * a[k]; }   // :146, :148
    if (TF) {
#pragma unroll
        for (int k = 0; k < D; ++k) out[k] = t[k];
        return;
    }
    check_to_odds<D>(t, out);                             // :147
#else
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        const double t = pre[k] * S;                      // :146
        out[k] = TF ? t : (1.0 - t) / (1.0 + t);          // :147
        S = S * a[k];                                     // :148
    }
#endif
}

template <int D, bool FIRST, bool TF = false>
__device__ __forceinline__ double check_update_exact(double *M, double sigma, double r, double S0 = 1.0)
{
    double a[D];
    if (FIRST) {
        const double a0 = 2.0 / (1.0 + r) - 1.0;          // every bit->check message is still r (:129)
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = a0;
    } else {
        double m[D];
#pragma unroll
        for (int k = 0; k < D; ++k) m[k] = ldm(M + (size_t)k * kTile);
        check_factors<D>(m, a);                                        // :140 / :148 (same value both times)
    }
    return check_finish_exact<D, TF>(M, a, sigma, S0);
}

// Two checks of degree D at once: all 2 D row loads are issued before anything is computed (twice the bytes in
// flight per wave; used where there are registers to spare: bp_team_kernels.hpp).  Same arithmetic per check.
template <int D, bool TF = false>
__device__ __forceinline__ void check_update_pair(double *M0, double *M1, double sigma0, double sigma1)
{
    double m0[D], m1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) m0[k] = ldm(M0 + (size_t)k * kTile);
#pragma unroll
    for (int k = 0; k < D; ++k) m1[k] = ldm(M1 + (size_t)k * kTile);
    loads_first();
    double a[D];
    check_factors<D>(m0, a);
    check_finish_exact<D, TF>(M0, a, sigma0);
    check_factors<D>(m1, a);
    check_finish_exact<D, TF>(M1, a, sigma1);
}

// Any degree, O(deg^2) recomputation of the prefix, still in place (position k is
// overwritten only after every prefix that needs it was formed).  Used above the widest
// straight-line variant.
template <bool FIRST>
__device__ __noinline__ void check_update_any(double *M, int deg, double sigma, double r)
{
    double S = 1.0;
    for (int k = deg - 1; k >= 0; --k) {
        double P = sigma;
        for (int q = 0; q < k; ++q) {
            const double m = FIRST ? r : M[(size_t)q * kTile];
            P = P * (2.0 / (1.0 + m) - 1.0);
        }
        const double mk = FIRST ? r : M[(size_t)k * kTile];
        const double ak = 2.0 / (1.0 + mk) - 1.0;
        const double t = P * S;
        M[(size_t)k * kTile] = (1.0 - t) / (1.0 + t);
        S = S * ak;
    }
}

// wave-uniform dispatch on the degree: deg in [LO, HI] -> check_update_exact<deg>
template <int LO, int HI, bool FIRST>
__device__ __forceinline__ double check_dispatch(double *M, int deg, double sigma, double r, double S0 = 1.0)
{
    if constexpr (LO == HI) {
        return check_update_exact<LO, FIRST>(M, sigma, r, S0);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (deg <= MID) return check_dispatch<LO, MID, FIRST>(M, deg, sigma, r, S0);
        return check_dispatch<MID + 1, HI, FIRST>(M, deg, sigma, r, S0);
    }
}

// A check of DC < deg <= 2 DC edges as two halves of straight-line code instead of the O(deg^2) path: the prefix
// product over the first DC edges (nothing stored), then the other deg - DC edges as a check of their own that starts
// from that prefix and hands back its suffix product, then the first DC edges once more (their rows are in the L2 by
// now) with the suffix starting there.  The same multiplications on the same operands in the same order as one pass
// over all deg edges (:136-148) -- the first half's factors are simply formed twice.
template <int DC, bool FIRST>
__device__ __forceinline__ void check_update_halves(double *M, int deg, double sigma, double r)
{
    double a[DC];
    auto factors = [&]() {
        if (FIRST) {
            const double a0 = 2.0 / (1.0 + r) - 1.0;
#pragma unroll
            for (int k = 0; k < DC; ++k) a[k] = a0;
        } else {
            double m[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k) m[k] = ldm(M + (size_t)k * kTile);
            check_factors<DC>(m, a);
        }
    };
    factors();
    double P = sigma;
#pragma unroll
    for (int k = 0; k < DC; ++k) P = P * a[k];                          // :139-140
    const double S = check_dispatch<1, DC, FIRST>(M + (size_t)DC * kTile, deg - DC, P, r);
    factors();
    check_finish_exact<DC>(M, a, sigma, S);
}

// HALVES: checks of up to 2 DC edges in two halves (instantiations whose widest bucket is narrower than their graphs'
// widest checks: bp_team_kernels.hpp, IRR)
template <int DC, bool FIRST, bool HALVES = false>
__device__ __forceinline__ void check_update(double *M, int deg, double sigma, double r)
{
    if (deg == DC) check_update_exact<DC, FIRST>(M, sigma, r);   // the regular-code case first
    else if (deg == 0) return;
    else if (deg < DC) check_dispatch<1, DC - 1, FIRST>(M, deg, sigma, r);
    else if (HALVES && deg <= 2 * DC) check_update_halves<DC, FIRST>(M, deg, sigma, r);
    else check_update_any<FIRST>(M, deg, sigma, r);
}

// ---------------------------------------------------------------------------
// variable-node update for one bit and 64 syndromes (one wave).
// belief_propagation.jl:153-177.  Returns the posterior odds T.
// `Mt` = tile message base + lane; pos = CSR positions of the bit's edges.
// ---------------------------------------------------------------------------
// the arithmetic of a bit once its D incoming messages c[k] (rows at[k]) are loaded
template <int D>
__device__ __forceinline__ double bit_finish_exact(double *Mt, const size_t (&at)[D], const double (&c)[D], double r)
{
    double pre[D];
    double F = r;                                         // :153
#pragma unroll
    for (int k = 0; k < D; ++k) {
        pre[k] = F;                                       // :156
        F = F * c[k];                                     // :157
        if (F != F) F = 1.0;                              // :158-160
    }
    double G = 1.0;                                       // :170
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        stm(Mt + at[k], pre[k] * G);                      // :172 (unguarded, may be NaN)
        G = G * c[k];                                     // :173
        if (G != G) G = 1.0;                              // :174-176
    }
    return F;
}

// ... the same with the D new messages handed back instead of stored
template <int D>
__device__ __forceinline__ double bit_compute_exact(const double (&c)[D], double r, double (&out)[D])
{
    double pre[D];
    double F = r;                                         // :153
#pragma unroll
    for (int k = 0; k < D; ++k) {
        pre[k] = F;                                       // :156
        F = F * c[k];                                     // :157
        if (F != F) F = 1.0;                              // :158-160
    }
    double G = 1.0;                                       // :170
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        out[k] = pre[k] * G;                              // :172 (unguarded, may be NaN)
        G = G * c[k];                                     // :173
        if (G != G) G = 1.0;                              // :174-176
    }
    return F;
}

// (pos as values already in registers; TF: the rows hold the check sweep's products t, see check_finish_exact)
template <int D, bool TF = false>
__device__ __forceinline__ double bit_update_exact_v(double *Mt, const int (&pos)[D], double r)
{
    double c[D];
    size_t at[D];
#pragma unroll
    for (int k = 0; k < D; ++k) at[k] = (size_t)pos[k] * kTile;
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = ldm(Mt + at[k]);
    if (TF) check_to_odds<D>(c, c);   // :147
    return bit_finish_exact<D>(Mt, at, c, r);
}

template <int D, bool TF = false>
__device__ __forceinline__ void bit_update_pair_v(double *Mt, const int (&pos0)[D], const int (&pos1)[D], double r, double &T0, double &T1)
{
    double c0[D], c1[D];
    size_t at0[D], at1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { at0[k] = (size_t)pos0[k] * kTile; at1[k] = (size_t)pos1[k] * kTile; }
#pragma unroll
    for (int k = 0; k < D; ++k) c0[k] = ldm(Mt + at0[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) c1[k] = ldm(Mt + at1[k]);
    loads_first();
    if (TF) { check_to_odds<D>(c0, c0); check_to_odds<D>(c1, c1); }   // :147
    T0 = bit_finish_exact<D>(Mt, at0, c0, r);
    T1 = bit_finish_exact<D>(Mt, at1, c1, r);
}

template <int D>
__device__ __forceinline__ double bit_update_exact(double *Mt, const int *__restrict__ pos, double r)
{
    double c[D];
    size_t at[D];
#pragma unroll
    for (int k = 0; k < D; ++k) at[k] = (size_t)pos[k] * kTile;
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = ldm(Mt + at[k]);
    return bit_finish_exact<D>(Mt, at, c, r);
}

// Two bits of degree D at once: every row load is issued before anything is computed.  Same arithmetic per bit;
// T0 / T1 are the posterior odds.
template <int D>
__device__ __forceinline__ void bit_update_pair(double *Mt, const int *__restrict__ pos0, const int *__restrict__ pos1, double r, double &T0, double &T1)
{
    double c0[D], c1[D];
    size_t at0[D], at1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { at0[k] = (size_t)pos0[k] * kTile; at1[k] = (size_t)pos1[k] * kTile; }
#pragma unroll
    for (int k = 0; k < D; ++k) c0[k] = ldm(Mt + at0[k]);
#pragma unroll
    for (int k = 0; k < D; ++k) c1[k] = ldm(Mt + at1[k]);
    T0 = bit_finish_exact<D>(Mt, at0, c0, r);
    T1 = bit_finish_exact<D>(Mt, at1, c1, r);
}

__device__ __noinline__ double bit_update_any(double *Mt, const int *__restrict__ pos, int deg, double r)
{
    double F = r;
    for (int k = 0; k < deg; ++k) {
        F = F * Mt[(size_t)pos[k] * kTile];
        if (F != F) F = 1.0;
    }
    double G = 1.0;
    for (int k = deg - 1; k >= 0; --k) {
        double Pk = r;
        for (int q = 0; q < k; ++q) {
            Pk = Pk * Mt[(size_t)pos[q] * kTile];
            if (Pk != Pk) Pk = 1.0;
        }
        const size_t a = (size_t)pos[k] * kTile;
        const double ck = Mt[a];
        Mt[a] = Pk * G;
        G = G * ck;
        if (G != G) G = 1.0;
    }
    return F;
}

template <int LO, int HI>
__device__ __forceinline__ double bit_dispatch(double *Mt, const int *__restrict__ pos, int deg, double r)
{
    if constexpr (LO == HI) {
        return bit_update_exact<LO>(Mt, pos, r);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (deg <= MID) return bit_dispatch<LO, MID>(Mt, pos, deg, r);
        return bit_dispatch<MID + 1, HI>(Mt, pos, deg, r);
    }
}

template <int DV>
__device__ __forceinline__ double bit_update(double *Mt, const int *__restrict__ pos, int deg, double r)
{
    if (deg == DV) return bit_update_exact<DV>(Mt, pos, r);
    if (deg == 0) return r;
    if (deg < DV) return bit_dispatch<1, DV - 1>(Mt, pos, deg, r);
    return bit_update_any(Mt, pos, deg, r);
}

// ---------------------------------------------------------------------------
// Straggler hand-off helpers (shared with bp_team_kernels.hpp).
// ---------------------------------------------------------------------------
// Room for cnt more syndromes in the next level?  Returns the first position, or ~0u when the level is full
// (compare-and-swap, so the count never runs past the capacity and a refused tile leaves no trace).
__device__ __forceinline__ unsigned int defer_reserve(unsigned int *count, unsigned int cnt, unsigned int cap)
{
    unsigned int old = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        if (old + cnt > cap || old + cnt < old) return ~0u;
        const unsigned int prev = atomicCAS(count, old, old + cnt);
        if (prev == old) return old;
        old = prev;
    }
}

// Copy this lane's column of the message rows first, first + step, ... (< nnz) of a tile into its place in the
// next level's packed tile (dst = packed tile base + lane-in-packed-tile; only lanes with `mine` store).  Whole
// 512-byte rows are read (the active lanes are scattered over the row), at most 16 x 8 contiguous bytes written.
__device__ __forceinline__ void defer_copy_rows(const double *Mt, double *dst, bool mine, int nnz, int first, int step)
{
    int e = first;
    for (; e + 7 * step < nnz; e += 8 * step) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = Mt[(size_t)(e + k * step) * kTile];
        if (mine) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[(size_t)(e + k * step) * kTile] = v[k];
        }
    }
    for (; e < nnz; e += step) {
        const double v = Mt[(size_t)e * kTile];
        if (mine) dst[(size_t)e * kTile] = v;
    }
}

// ---------------------------------------------------------------------------
// The message-passing kernel: persistent workgroups, one tile at a time.
// ---------------------------------------------------------------------------
// The Tanner graph (int32, device resident, read-only for the whole launch) and the
// packed syndromes are separate __restrict__ arguments so that their wave-uniform reads
// become scalar loads:
//   row_ptr  [s+1]  CSR (check-major) edge ranges
//   edge_bit [nnz]  bit of every CSR edge, ascending inside a check
//   col_ptr  [n+1]  CSC (bit-major) edge ranges
//   csc2csr  [nnz]  CSR position of every CSC edge, checks ascending inside a bit
//   synmask  [ntiles][s], nevermask [ntiles] (lanes holding a syndrome entry other than 0/1)
// SECOND = a pass over a packed level of handed-off syndromes (same code; a distinct instantiation only so that
// profilers list the passes as separate kernels instead of averaging a ~1 s launch with a ~5 us one).
// A lane of such a pass has already run it0 iterations; its messages are in its packed tile, and it counts on
// from there: every lane carries its own iteration total and retires (unconverged) when that reaches max_iters.
template <int DC, int DV, bool WANT_LLR, int THREADS, bool SECOND>
__global__ void
__launch_bounds__(THREADS, (min_waves_per_simd<DC, DV, THREADS>()))
bp_tile_kernel(BPParams p, const int *__restrict__ row_ptr, const int *__restrict__ edge_bit,
               const int *__restrict__ col_ptr, const int *__restrict__ csc2csr,
               const u64 *__restrict__ synmask, const u64 *__restrict__ nevermask)
{
    __shared__ int sh_tile;
    __shared__ unsigned int sh_base;
    __shared__ u64 sh_mism[THREADS / 64];
    const int lane = threadIdx.x & 63;
    // the wave index is wave-uniform; telling the compiler so turns every per-node index read
    // into a scalar load and every degree test into a scalar branch
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int W = THREADS / 64;
    const int s = p.s, n = p.n;
    const double r = p.r;
#if LDPC_ROTATE
    const int rot_c = s > 0 ? (int)((blockIdx.x * 2654435761u) % (unsigned)s) : 0;
    const int rot_v = n > 0 ? (int)((blockIdx.x * 2246822519u) % (unsigned)n) : 0;
#endif

    long long batch_ = p.batch;
    if (p.count_dev) {
        batch_ = (long long)*p.count_dev;
        if (batch_ <= (long long)p.count_skip) batch_ = 0;
    }
    const long long batch = batch_;
    const int ntiles = p.count_dev ? (int)((batch + kTile - 1) / kTile) : p.ntiles;

    for (;;) {
        if (threadIdx.x == 0) sh_tile = (int)atomicAdd(p.queue, 1u);
        __syncthreads();
        const int tile = sh_tile;
        if (tile >= ntiles) break;  // every wave of every workgroup reaches this

        // fresh tiles: the workgroup's own slot; a packed level (SECOND): the tile's packed tile, in place
        double *const Mt = p.msg + (size_t)(SECOND ? (unsigned)tile : blockIdx.x) * (size_t)p.slot_stride + lane;
        const u64 *syn = synmask + (size_t)tile * s;
        u64 *em = p.errmask + (size_t)tile * n;
        u64 *fin = p.finmask + (size_t)tile * n;
        const long long b0 = (long long)tile * kTile;
        const long long left = batch - b0;
        u64 deferred = 0;
        const u64 valid = left >= kTile ? ~0ull : ((1ull << left) - 1ull);
        const u64 never = nevermask[tile];
        u64 active = valid;
        // iterations this lane's syndrome has behind it (a resumed syndrome: its messages are already in Mt)
        const bool resumed = SECOND && p.resumed;
        const int it0 = (resumed && ((valid >> lane) & 1ull)) ? p.cold->it0[b0 + lane] : 0;
        int my_iters = 0;
        int my_conv = 0;
        int it = 0;
        u64 tk_check = 0, tk_var = 0, tk_conv = 0;

        while (active != 0) {   // (every lane retires at the latest when its total reaches max_iters)
            ++it;
            const bool first = (it == 1) && !resumed;
            const u64 t0 = LDPC_PHASE_TICKS ? wall_clock64() : 0;
            // ---- check-node sweep  (:135-150)
            for (int i0 = w; i0 < s; i0 += W) {
#if LDPC_ROTATE
                const int i = (i0 + rot_c >= s) ? i0 + rot_c - s : i0 + rot_c;
#else
                const int i = i0;
#endif
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const double sigma = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0;  // (-1)^syndrome[i] :136
                if (first) check_update<DC, true>(Mt + (size_t)e0 * kTile, deg, sigma, r);
                else check_update<DC, false>(Mt + (size_t)e0 * kTile, deg, sigma, r);
            }
            __syncthreads();
            const u64 t1 = LDPC_PHASE_TICKS ? wall_clock64() : 0;
            // ---- variable-node sweep  (:152-178)
            auto finish_bit = [&](int j, double T) {
                const u64 dec = __ballot(T >= 1.0);                            // :164-168
                if (WANT_LLR) {
                    if ((active >> lane) & 1ull)
                        p.llr[((size_t)tile * n + j) * kTile + lane] = llr_of(T, p.llr_exact);  // :163
                }
                if (lane == 0) em[j] = dec;   // every lane, also the stopped ones: theirs were captured when they stopped
            };
            auto rotated_bit = [&](int j0) {
#if LDPC_ROTATE
                return (j0 + rot_v >= n) ? j0 + rot_v - n : j0 + rot_v;
#else
                return j0;
#endif
            };
            // (requesting the rows of the NEXT bit before this one is finished -- 8 rows in flight per wave instead of
            // 4 behind two scalar index loads -- was built and measured in round 2: the variable sweep's share fell
            // from 59 % to 49 %, the check sweep's rose by as much, and the launch took 1156 instead of 1158 ms: the
            // three workgroups of a CU sit in different phases and the memory system is what they share.  Removed.)
            for (int j0 = w; j0 < n; j0 += W) {
                const int j = rotated_bit(j0);
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                finish_bit(j, bit_update<DV>(Mt, csc2csr + c0, deg, r));
            }
            __syncthreads();
            const u64 t2 = LDPC_PHASE_TICKS ? wall_clock64() : 0;
            // ---- convergence test (:180-184): lane = check, words = 64 syndromes
            u64 mism = 0;
            for (int i = w * 64 + lane; i < s; i += W * 64) {
                u64 par = 0;
                const int e1 = row_ptr[i + 1];
                // 8 independent index loads, then 8 independent word gathers per step
                // (a plain loop serialises 2*deg dependent L2 round trips)
                for (int e = row_ptr[i]; e < e1; e += 8) {
                    int jb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) jb[q] = (e + q < e1) ? edge_bit[e + q] : -1;
                    u64 wv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) wv[q] = (jb[q] >= 0) ? em[jb[q]] : 0ull;
#pragma unroll
                    for (int q = 0; q < 8; ++q) par ^= wv[q];
                }
                mism |= par ^ syn[i];
            }
            mism = wave_or(mism);
            if (lane == 0) sh_mism[w] = mism;
            __syncthreads();
            u64 U = never;
#pragma unroll
            for (int q = 0; q < W; ++q) U |= sh_mism[q];
            U = uniform64(U);   // the same in every lane: keep it, and the lane masks derived from it, in scalar registers
            const int total = it0 + it;                      // iterations of this lane's syndrome so far
            const u64 newly = active & ~U;
            if ((newly >> lane) & 1ull) { my_iters = total; my_conv = 1; }
            active &= U;
            // out of iterations (belief_propagation.jl:134): the lane retires unconverged with the decisions /
            // LLRs of this, its last, iteration
            const u64 spent = __ballot(total >= p.max_iters) & active;
            if ((spent >> lane) & 1ull) { my_iters = total; my_conv = 0; }
            active &= ~spent;
            // the lanes that stop here keep the decisions of THIS iteration (the reference breaks at :183 / leaves the
            // loop at :134): capture their columns (lane = bit, 64 words per wave step)
            const u64 stopped = newly | spent;
            if (stopped != 0) {
                for (int j = w * 64 + lane; j < n; j += W * 64) fin[j] = (fin[j] & ~stopped) | (em[j] & stopped);
            }
            const u64 t3 = LDPC_PHASE_TICKS ? wall_clock64() : 0;
            tk_check += t1 - t0; tk_var += t2 - t1; tk_conv += t3 - t2;
            // few stragglers left: hand them -- with their messages -- to the next level instead of sweeping a
            // nearly empty tile (uniform over the workgroup: active and it are)
            if (p.defer_thresh != 0 && active != 0 && it >= p.defer_min_iter && (int)__popcll(active) <= p.defer_thresh) {
                const BPCold *const cd = p.cold;
                if (threadIdx.x == 0) sh_base = defer_reserve(cd->defer_count, (unsigned)__popcll(active), cd->next_cap);
                __syncthreads();
                const unsigned base = sh_base;
                if (base != ~0u) {
                    const bool mine = (active >> lane) & 1ull;
                    const unsigned q = base + __builtin_amdgcn_mbcnt_hi((unsigned)(active >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)active, 0u));   // + active lanes below this one
                    defer_copy_rows(Mt, cd->next_state + (size_t)(q >> 6) * (size_t)cd->next_stride + (q & 63u), mine, p.nnz, w, W);
                    if (w == 0 && mine) {
                        cd->defer_list[q] = resumed ? cd->index[b0 + lane] : (int)(b0 + lane);
                        cd->defer_it[q] = total;
                    }
                    deferred = active;
                    active = 0;
                }
            }
        }
        if (w == 0) {
            const BPCold *const cd = p.cold;
            if (((valid & ~deferred) >> lane) & 1ull) {
                const long long ob = resumed ? (long long)cd->index[b0 + lane] : b0 + lane;
                cd->conv[ob] = (unsigned char)my_conv;
                int *const iters = cd->iters;
                if (iters) iters[ob] = my_iters;
            } else {
                my_iters = 0;
            }
            // sum of iterations executed, for the algorithmic byte count (a handed-off syndrome is counted by the
            // pass that finishes it, with its total: every iteration is executed, and counted, once)
            int tot = my_iters;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
            if (lane == 0) {
                atomicAdd(cd->sum_iters, (u64)tot);
                if (LDPC_PHASE_TICKS) {
                    u64 *const ticks = cd->phase_ticks;
                    atomicAdd(&ticks[0], tk_check);
                    atomicAdd(&ticks[1], tk_var);
                    atomicAdd(&ticks[2], tk_conv);
                }
            }
        }
        __syncthreads();
    }
}

#ifdef LDPC_AUX_KERNELS   // the small non-template kernels: only the host translation unit (ldpc_mi355x.hip) defines them
// the per-pass BPCold blocks of one call: by-value kernel arguments -> device memory, in stream order
__global__ void __launch_bounds__(64) store_cold_kernel(BPCold c0, BPCold c1, BPCold c2, BPCold *dst)
{
    if (threadIdx.x == 0) { dst[0] = c0; dst[1] = c1; dst[2] = c2; }
}

// include/ldpc_mi355x_debug.h ldpc_debug_div_check: div_core against the IEEE division
__global__ void __launch_bounds__(256) div_check_kernel(const double *num, const double *den, double *out_core, double *out_ieee, long long count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    out_core[i] = div_core(num[i], den[i]);
    out_ieee[i] = num[i] / den[i];
}

// include/ldpc_mi355x_debug.h ldpc_debug_llr_check: the LLR every kernel returns (llr_of: llr_cut of the cut odds) against
// the library's log(1 / .) of the same cut odds
__global__ void __launch_bounds__(256) llr_check_kernel(const double *T, double *out_fast, double *out_lib, long long count)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    out_fast[i] = llr_of(T[i], 0);
    out_lib[i] = log(1.0 / llr_from_hi32(llr_hi32(T[i])));
}

// ---------------------------------------------------------------------------
// placement probe: the two sweeps' access patterns on a candidate workspace, with the real kernel's geometry (one
// 8-wave workgroup per slot, three per CU, every workgroup starting at a rotation of its own) -- every wave gathers
// 4 pseudo-random rows of the slot and writes them back (variable sweep), then streams 8 contiguous rows in place
// (check sweep).  The host times it on candidate chunk groups and keeps the fastest (DESIGN.md "Workspace
// placement").  Traffic: 4 x rows x 512 B per slot.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 6) placement_probe_kernel(double *base, long long slot_stride, int rows)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *M = base + (size_t)blockIdx.x * (size_t)slot_stride + lane;
    const unsigned r = (unsigned)rows, nb = r / 4u, nc = r / 8u;
    const unsigned rot_v = nb ? (blockIdx.x * 2246822519u) % nb : 0u, rot_c = nc ? (blockIdx.x * 2654435761u) % nc : 0u;
    for (unsigned j0 = (unsigned)w; j0 < nb; j0 += 8u) {
        const unsigned j = (j0 + rot_v >= nb) ? j0 + rot_v - nb : j0 + rot_v;
        const unsigned a = (j * 2654435761u + 12345u) % r, b = (j * 2246822519u + 977u) % r,
                       c = (j * 3266489917u + 31u) % r, d = (j * 668265263u + 7u) % r;
        const double v0 = M[(size_t)a * kTile], v1 = M[(size_t)b * kTile], v2 = M[(size_t)c * kTile],
                     v3 = M[(size_t)d * kTile];
        M[(size_t)a * kTile] = v1; M[(size_t)b * kTile] = v2; M[(size_t)c * kTile] = v3; M[(size_t)d * kTile] = v0;
    }
    __syncthreads();
    for (unsigned i0 = (unsigned)w; i0 < nc; i0 += 8u) {
        const unsigned i = (i0 + rot_c >= nc) ? i0 + rot_c - nc : i0 + rot_c;
        double *R = M + (size_t)i * 8 * kTile;
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = R[(size_t)k * kTile];
#pragma unroll
        for (int k = 0; k < 8; ++k) R[(size_t)k * kTile] = v[k] * 1.0000001;   // (a store of the loaded value would be elided)
    }
}

// ---------------------------------------------------------------------------
// pack: syndromes uint8 [batch][s]  ->  lane masks synmask[tile][s] (+ nevermask)
// one wave per (tile, 64 checks); lane = check.
// ---------------------------------------------------------------------------
// index / count_dev (both nullptr in the first pass): compact syndrome q of the second pass is
// syndrome index[q] of the batch, and there are *count_dev of them -- unless that is at most
// count_skip, in which case the tile-kernel second pass does not run at all (the node-parallel
// kernel decodes those few syndromes straight from / into the caller's arrays).
// V = 4: a lane takes four consecutive checks (bits) -- one 4-byte load (store) per syndrome instead of four single bytes, a
// wave 256 contiguous bytes per instruction instead of 64 --; the host picks it when s (n) is a multiple of 4 and the
// caller's array is 4-byte aligned, V = 1 otherwise.
template <int V>
__global__ void __launch_bounds__(64) pack_syndromes_kernel(const unsigned char *syn, long long batch,
                                                            int s, u64 *synmask, u64 *nevermask,
                                                            const int *index, const unsigned int *count_dev,
                                                            unsigned int count_skip)
{
    static_assert(V == 1 || V == 4, "one check or four per lane");
    if (count_dev) { batch = (long long)*count_dev; if (batch <= (long long)count_skip) return; }
    const int tile = blockIdx.y;
    const int i = (blockIdx.x * 64 + threadIdx.x) * V;
    const long long b0 = (long long)tile * kTile;
    if (b0 >= batch) return;
    const int rows = (int)((batch - b0) < kTile ? (batch - b0) : kTile);
    u64 m[V], hi = 0;
#pragma unroll
    for (int q = 0; q < V; ++q) m[q] = 0;
    if (i < s) {
#pragma unroll 8
        for (int rr = 0; rr < rows; ++rr) {
            const long long b = index ? (long long)index[b0 + rr] : b0 + rr;
            if constexpr (V == 4) {
                const unsigned v = *(const unsigned *)(syn + (size_t)b * s + i);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned c = (v >> (8 * q)) & 0xffu;
                    m[q] |= (u64)(c & 1u) << rr;
                    hi |= (u64)(c > 1u) << rr;
                }
            } else {
                const unsigned v = syn[(size_t)b * s + i];
                m[0] |= (u64)(v & 1u) << rr;
                hi |= (u64)(v > 1u) << rr;
            }
        }
#pragma unroll
        for (int q = 0; q < V; ++q) synmask[(size_t)tile * s + i + q] = m[q];
    }
    hi = wave_or(hi);
    if (threadIdx.x == 0 && hi) atomicOr(&nevermask[tile], hi);
}

// unpack: errmask[tile][n] -> errors uint8 [batch][n]; one wave per (tile, 64 V bits)
template <int V>
__global__ void __launch_bounds__(64) unpack_errors_kernel(const u64 *errmask, long long batch, int n,
                                                           unsigned char *errors, const int *index,
                                                           const unsigned int *count_dev, unsigned int count_skip)
{
    static_assert(V == 1 || V == 4, "one bit or four per lane");
    if (count_dev) { batch = (long long)*count_dev; if (batch <= (long long)count_skip) return; }
    const int tile = blockIdx.y;
    const int j = (blockIdx.x * 64 + threadIdx.x) * V;
    const long long b0 = (long long)tile * kTile;
    if (b0 >= batch) return;
    const int rows = (int)((batch - b0) < kTile ? (batch - b0) : kTile);
    if (j >= n) return;
    u64 m[V];
#pragma unroll
    for (int q = 0; q < V; ++q) m[q] = errmask[(size_t)tile * n + j + q];
#pragma unroll 8
    for (int rr = 0; rr < rows; ++rr) {
        const long long b = index ? (long long)index[b0 + rr] : b0 + rr;
        if constexpr (V == 4) {
            const unsigned v = (unsigned)((m[0] >> rr) & 1ull) | (unsigned)((m[1] >> rr) & 1ull) << 8 |
                               (unsigned)((m[2] >> rr) & 1ull) << 16 | (unsigned)((m[3] >> rr) & 1ull) << 24;
            *(unsigned *)(errors + (size_t)b * n + j) = v;
        } else {
            errors[(size_t)b * n + j] = (unsigned char)((m[0] >> rr) & 1ull);
        }
    }
}

// llr transpose: llr_t[tile][n][64] -> llr[batch][n]; 64x64 tile through LDS.  raw = 4 / 5: llr_t holds the posterior odds T
// as the team kernel's fresh pass left them (TeamParams::llr_raw: upper 32 bits / all of T, position-chunk layout) and
// log(1 / T) (:163) is taken here, once per syndrome and bit.  All 16 loads of a thread are issued before anything is
// computed (one load after the other behind its own index load, as this kernel first was, took 7.2 ms for a batch of
// 65,536 x 16384: the latency of 2 x 16 dependent loads per workgroup, not bytes).
#ifndef LDPC_UNPACK_XCD
#define LDPC_UNPACK_XCD 1
#endif
__global__ void __launch_bounds__(256) unpack_llr_kernel(const double *llr_t, long long batch, int n,
                                                         double *llr, const int *index,
                                                         const unsigned int *count_dev, unsigned int count_skip, int raw, int exact,
                                                         const int *posmap)
{
    __shared__ double t[64][65];
    __shared__ int spos[64];
    if (count_dev) { batch = (long long)*count_dev; if (batch <= (long long)count_skip) return; }   // uniform
    // Workgroups go round-robin over the 8 XCDs in launch order: all the workgroups of a tile are given the same residue,
    // so that one L2 sees the tile's scratch rows -- in the position-chunk layout a wave's 4-byte gather uses a quarter
    // of every line it touches, and the other three quarters belong to workgroups of the same tile.  (gridDim.y is a
    // multiple of 8: tiles beyond the batch leave below.)
#if LDPC_UNPACK_XCD
    const unsigned int lin = blockIdx.y * gridDim.x + blockIdx.x, per8 = 8u * gridDim.x;
    const int tile = (int)((lin / per8) * 8u + (lin % per8) % 8u);
    const int j0 = (int)((lin % per8) / 8u) * 64;
#else
    const int tile = blockIdx.y;
    const int j0 = blockIdx.x * 64;
#endif
    const long long b0 = (long long)tile * kTile;
    if (b0 >= batch) return;   // uniform for the whole workgroup
    const int rows = (int)((batch - b0) < kTile ? (batch - b0) : kTile);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (raw == 4 || raw == 5) {
        // the team kernel's scratch layout (bp_team_kernels.hpp): element (position, lane) of a tile at
        // ((position / 4) * 64 + lane) * 4 + position % 4, rows rounded up to a multiple of 4; posmap = the
        // position of every bit in the dealt order (nullptr: the bit itself)
        if (threadIdx.x < 64) spos[tx] = j0 + tx < n ? (posmap ? posmap[j0 + tx] : j0 + tx) : -1;
        __syncthreads();
        const size_t base = (size_t)tile * (((size_t)n + 3) & ~(size_t)3) * kTile + (size_t)tx * 4;
        if (raw == 4) {
            const unsigned int *const src = (const unsigned int *)llr_t + base;
            unsigned int u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pos = spos[ty + 4 * q];
                u[q] = pos >= 0 ? src[(size_t)(pos >> 2) * (kTile * 4) + (size_t)(pos & 3)] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) t[ty + 4 * q][tx] = llr_cut(llr_from_hi32(u[q]));
        } else {
            const double *const src = llr_t + base;
            double u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int pos = spos[ty + 4 * q];
                u[q] = pos >= 0 ? src[(size_t)(pos >> 2) * (kTile * 4) + (size_t)(pos & 3)] : 1.0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) t[ty + 4 * q][tx] = llr_of(u[q], exact);
        }
    } else {
        double u[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int jj = ty + 4 * q;
            u[q] = j0 + jj < n ? llr_t[((size_t)tile * n + j0 + jj) * kTile + tx] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) t[ty + 4 * q][tx] = u[q];
    }
    __syncthreads();
    if (j0 + tx >= n) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int rr = ty + 4 * q;
        if (rr < rows) {
            const long long b = index ? (long long)index[b0 + rr] : b0 + rr;
            llr[(size_t)b * n + j0 + tx] = t[tx][rr];
        }
    }
}

#endif  // LDPC_AUX_KERNELS

}  // namespace ldpc
