// bp_team_kernels.hpp -- medium batches on Tanner graphs beyond the LDS: several workgroups per tile.
//
// The tile kernel (bp_kernels.hpp) gives one 64-syndrome tile to ONE workgroup: a batch of a few
// thousand syndromes of the n = 16384 code is 32 ... 256 tiles, each swept by one CU at one CU's
// pace (~50 GB/s: 2.7 ms per iteration) while the HBM idles.  Here a TEAM of G workgroups shares a
// tile -- same layout (msg[edge][64] in the tile's workspace slot, lane = syndrome), same node
// updates (check_update / bit_update of bp_kernels.hpp, so the results are bit-identical), the
// nodes of every sweep dealt round-robin over the G x W waves of the team -- and the workgroup
// barriers between the sweeps become team barriers:
//
//   every storing wave: s_waitcnt vmcnt(0); workgroup barrier; one lane: agent-scope RELEASE
//   (buffer_wbl2 sc1), s_waitcnt vmcnt(0), relaxed agent-scope add to the team's arrival counter,
//   sc1-load poll until released, agent-scope ACQUIRE (buffer_inv sc1), s_waitcnt vmcnt(0);
//   workgroup barrier; plain loads             (MI355X_MICROARCH.md, inter-workgroup visibility)
//
// because per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another
// CU's stores.  All G workgroups of a team must be resident at once: the host launches
// ntiles x G <= 2 workgroups per CU with a cooperative launch (checked against the kernel's residency,
// never two cooperative grids at once) and every poll is bounded --
// a team that waits longer than ~10 s raises a fault word (host-mapped, reported by the next call on
// the handle) and all its members leave, so a lost workgroup can never hang the GPU.
#pragma once
#include "bp_kernels.hpp"

#ifndef LDPC_TEAM_SLEEP   // s_sleep argument between two polls of the arrival counter (x64 cycles)
#define LDPC_TEAM_SLEEP 16
#endif

namespace ldpc {

struct TeamParams {
    int G;                      // workgroups per tile
    // per tile one control block of kTeamCtlWords words, zero at launch (see team_barrier)
    unsigned int *ctl;
    u64 *mism;                  // [ntiles][mism_stride] mismatch words of the convergence tests, zero at launch
    int mism_stride;            // >= max_iters, a multiple of 32 words
    unsigned int *fault;        // host-mapped: set when a team barrier timed out
    int always_release;         // experiments: 1 = release at every barrier even when the team shares one XCD
    int scatter;                // tests: 1 = deal a team's members over ALL XCDs (exercises the release path and
                                // the XCC check that selects it); gridDim.x == ntiles * G
    // As the second pass of the straggler hand-off (p.count_dev != nullptr): the number of syndromes is
    // only known on the device, so the geometry is worked out here -- G above is then the cap, and the
    // pass runs only for p.count_skip < *count_dev <= count_max (fewer: node kernel, more: packed tiles).
    unsigned int count_max;
    int inject_fault;           // tests: raise the fault word and leave at once, as if a team barrier had timed out
    unsigned int ticket;        // what a timed-out barrier writes into the fault word: the number of this call on its
                                // handle (never 0), so that the report can name the first call that was hit
};

// Team barrier number k (1, 2, ...).  Control block of a tile: arrival counter at word 0, XCC mask at word
// 32, then one 128-byte line per member holding the number of the last barrier it was released from.
// The last arriver (it knows from the value its add returned) writes k into every member's line; the
// others poll THEIR OWN line.  (All members polling the one counter cost O(G^2) memory requests per
// barrier -- every add drops the line from every poller's L2 -- and with 8 ... 16 teams that storm took
// three times as long as the sweeps themselves.)
// one_xcd: every member of the team runs on the same XCD, so their stores meet in ONE L2 and the
// release (a write-back of that L2) is not needed; the acquire (this CU's L1) always is.
// Returns false when the team is broken (somebody timed out): the caller leaves the kernel.
constexpr int kTeamCtlWords = 32 * (2 + 64);   // per tile: counter line, XCC line, up to 64 member lines

__device__ __forceinline__ bool team_barrier(unsigned int *ctl, int G, int rank, unsigned int k, unsigned int *fault,
                                             unsigned int ticket, int *sh_ok, bool one_xcd)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!one_xcd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the wait after buffer_wbl2
        }
        int ok = 1;
        const unsigned int prev = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == k * (unsigned)G) {
            for (int m = 0; m < G; ++m)
                __hip_atomic_store(ctl + 64 + 32 * m, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned int *mine = ctl + 64 + 32 * rank;
            const u64 t0 = wall_clock64();
            unsigned int polls = 0;
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
                __builtin_amdgcn_s_sleep(LDPC_TEAM_SLEEP);
                // the fault word lives in HOST memory (a poll of it is a PCIe read: 256 pollers doing that every
                // turn tripled the barrier time) -- look at it, and at the clock, once in 256 turns
                if ((++polls & 255u) == 0u) {
                    if (__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { ok = 0; break; }
                    if (wall_clock64() - t0 > 1000000000ull) {   // 10 s of the 100 MHz clock
                        __hip_atomic_store(fault, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        ok = 0;
                        break;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // holds the barrier until the invalidate is through
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}

template <int DC, int DV, bool WANT_LLR, int THREADS>
__global__ void
__launch_bounds__(THREADS, (min_waves_per_simd<DC, DV, THREADS>()))
bp_team_kernel(BPParams p, TeamParams tp, const int *__restrict__ row_ptr, const int *__restrict__ edge_bit,
               const int *__restrict__ col_ptr, const int *__restrict__ csc2csr,
               const u64 *__restrict__ synmask, const u64 *__restrict__ nevermask)
{
    __shared__ int sh_ok;
    __shared__ u64 sh_mism[THREADS / 64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int W = THREADS / 64;
    const int s = p.s, n = p.n;
    const double r = p.r;
    if (tp.inject_fault) {
        if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(tp.fault, tp.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    int G = tp.G, ntiles = p.ntiles;
    long long batch = p.batch;
    if (p.count_dev) {                                         // second pass: sized on the device
        batch = (long long)*p.count_dev;
        if (batch <= (long long)p.count_skip || batch > (long long)tp.count_max) return;
        ntiles = (int)((batch + kTile - 1) / kTile);
        G = min(G, (int)(gridDim.x >> 3) / ((ntiles + 7) / 8));
        if (G < 1) return;                                     // (count_max keeps this from happening)
    }
    // Workgroups are dealt round-robin over the 8 XCDs (observed, not promised): blocks b and b + 8 share
    // one.  Teams are formed among the blocks of one residue class, so that normally a team sits on ONE
    // XCD; the members check it (xccs) and fall back to full release / acquire barriers if it is not so.
    const int bq = (int)(blockIdx.x >> 3), xslot = (int)(blockIdx.x & 7u);   // gridDim.x == 8 * G * ceil(ntiles / 8)
    const int tile = tp.scatter ? (int)(blockIdx.x / (unsigned)G) : (bq / G) * 8 + xslot;
    const int rank = tp.scatter ? (int)(blockIdx.x % (unsigned)G) : bq % G;
    if (tile >= ntiles || (p.count_dev && bq >= G * ((ntiles + 7) / 8))) return;   // whole teams only: nobody waits for these
    const int gw = rank * W + w, GW = G * W;                   // this wave among the team's waves
    double *const Mt = p.msg + (size_t)tile * (size_t)p.slot_stride + lane;
    unsigned int *const ctr = tp.ctl + (size_t)tile * kTeamCtlWords;
    unsigned int *const xccs = ctr + 32;
    u64 *const mw = tp.mism + (size_t)tile * (size_t)tp.mism_stride;
    unsigned int epoch = 0;                                    // barriers passed
    __shared__ int sh_one_xcd;
    if (threadIdx.x == 0) {
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(xccs, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool one_xcd = false;                                      // known after the first (full) barrier

    const u64 *syn = synmask + (size_t)tile * s;
    u64 *em = p.errmask + (size_t)tile * n;
    u64 *fin = p.finmask + (size_t)tile * n;
    const long long b0 = (long long)tile * kTile;
    const long long left = batch - b0;
    u64 deferred = 0;
    const u64 valid = left >= kTile ? ~0ull : ((1ull << left) - 1ull);
    const u64 never = nevermask[tile];
    u64 active = valid;                                        // identical in every member: same inputs, same words
    // a pass over a packed level: every lane has it0 iterations behind it and its messages in the packed tile
    const bool resumed = p.resumed != 0;
    const BPCold *const cd = p.cold;
    const int it0 = (resumed && ((valid >> lane) & 1ull)) ? cd->it0[b0 + lane] : 0;
    int my_iters = 0, my_conv = 0, it = 0;
    u64 tk_check = 0, tk_var = 0, tk_rest = 0;   // this wave's own sweep time / everything else (waiting included)

    while (active != 0) {   // (every lane retires at the latest when its total reaches max_iters)
        ++it;
        const bool first = (it == 1) && !resumed;
        const u64 t0 = wall_clock64();
        // ---- check-node sweep  (:135-150)
        for (int i = gw; i < s; i += GW) {
            const int e0 = row_ptr[i];
            const int deg = row_ptr[i + 1] - e0;
            const double sigma = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0;
            if (first) check_update<DC, true>(Mt + (size_t)e0 * kTile, deg, sigma, r);
            else check_update<DC, false>(Mt + (size_t)e0 * kTile, deg, sigma, r);
        }
        const u64 t1 = wall_clock64();
        if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd)) return;
        if (first) {
            if (threadIdx.x == 0)
                sh_one_xcd = __popc(__hip_atomic_load(xccs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
            __syncthreads();
            one_xcd = sh_one_xcd != 0 && !tp.always_release;
        }
        const u64 t2 = wall_clock64();
        // ---- variable-node sweep  (:152-178); a wave takes 16 consecutive bits at a time, so that a
        //      128-byte line of decision words has one writer
        for (int jb0 = gw * 16; jb0 < n; jb0 += GW * 16)
        for (int j = jb0; j < min(jb0 + 16, n); ++j) {
            const int c0 = col_ptr[j];
            const int deg = col_ptr[j + 1] - c0;
            const double T = bit_update<DV>(Mt, csc2csr + c0, deg, r);
            const u64 dec = __ballot(T >= 1.0);                            // :164-168
            if (WANT_LLR) {
                if ((active >> lane) & 1ull) p.llr[((size_t)tile * n + j) * kTile + lane] = log(1.0 / T);  // :163
            }
            if (lane == 0) em[j] = dec;   // every lane; the stopped ones' decisions were captured when they stopped
        }
        const u64 t3 = wall_clock64();
        if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd)) return;
        // ---- convergence test (:180-184): lane = check, words = 64 syndromes; the team ORs into mw[it-1]
        u64 mism = 0;
        for (int i = gw * 64 + lane; i < s; i += GW * 64) {
            u64 par = 0;
            const int e1 = row_ptr[i + 1];
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
        if (threadIdx.x == 0) {
            u64 part = 0;
#pragma unroll
            for (int q = 0; q < W; ++q) part |= sh_mism[q];
            if (part) __hip_atomic_fetch_or(&mw[it - 1], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd)) return;
        const u64 U = uniform64(never | __hip_atomic_load(&mw[it - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int total = it0 + it;                            // iterations of this lane's syndrome so far
        const u64 newly = active & ~U;
        if ((newly >> lane) & 1ull) { my_iters = total; my_conv = 1; }
        active &= U;
        const u64 spent = __ballot(total >= p.max_iters) & active;   // out of iterations: retires unconverged
        if ((spent >> lane) & 1ull) { my_iters = total; my_conv = 0; }
        active &= ~spent;
        // capture the stopping lanes' decisions of this iteration (every member its share of the bits; the decision
        // words were made visible by the barrier before the test, and nobody rewrites them before the barrier
        // after the next check sweep, which this member only joins when it is through here)
        const u64 stopped = newly | spent;
        if (stopped != 0) {
            for (int j = gw * 64 + lane; j < n; j += GW * 64) fin[j] = (fin[j] & ~stopped) | (em[j] & stopped);
        }
        const u64 t4 = wall_clock64();
        tk_check += t1 - t0; tk_var += t3 - t2; tk_rest += (t2 - t1) + (t4 - t3);   // rest = barriers + test
        // few stragglers left: hand them, with their messages, to the next level (decided alike by every member;
        // rank 0 reserves the room and tells the others through the tile's control block, then all copy)
        if (p.defer_thresh != 0 && active != 0 && it >= p.defer_min_iter && (int)__popcll(active) <= p.defer_thresh) {
            if (rank == 0 && threadIdx.x == 0)
                __hip_atomic_store(ctr + 33, defer_reserve(cd->defer_count, (unsigned)__popcll(active), cd->next_cap) + 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // base + 1; a full level (~0u) is told as 0
            if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd)) return;
            const unsigned told = __hip_atomic_load(ctr + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd)) return;   // everyone has read it: the word may be rewritten
            if (told != 0u) {                                  // (0 = ~0u + 1: the next level is full, carry on)
                const unsigned base = told - 1u;
                const bool mine = (active >> lane) & 1ull;
                const unsigned q = base + __builtin_amdgcn_mbcnt_hi((unsigned)(active >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)active, 0u));   // + active lanes below this one
                defer_copy_rows(Mt, cd->next_state + (size_t)(q >> 6) * (size_t)cd->next_stride + (q & 63u), mine, p.nnz, gw, GW);
                if (rank == 0 && w == 0 && mine) {
                    cd->defer_list[q] = resumed ? cd->index[b0 + lane] : (int)(b0 + lane);
                    cd->defer_it[q] = total;
                }
                deferred = active;
                active = 0;
            }
        }
    }
    if (rank == 0 && w == 0) {
        if (((valid & ~deferred) >> lane) & 1ull) {
            const long long ob = resumed ? (long long)cd->index[b0 + lane] : b0 + lane;
            cd->conv[ob] = (unsigned char)my_conv;
            if (cd->iters) cd->iters[ob] = my_iters;
        } else {
            my_iters = 0;
        }
        int tot = my_iters;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
        if (lane == 0) {
            atomicAdd(cd->sum_iters, (u64)tot);
            atomicAdd(&cd->phase_ticks[0], tk_check);
            atomicAdd(&cd->phase_ticks[1], tk_var);
            atomicAdd(&cd->phase_ticks[2], tk_rest);
        }
    }
}

}  // namespace ldpc
