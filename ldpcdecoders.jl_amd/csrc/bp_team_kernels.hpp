// bp_team_kernels.hpp -- Tanner graphs beyond the LDS: several workgroups per tile.  Built in round 1 for medium
// batches; with persistent teams (below) the kernel of the full C3 / C4 batches and of the headline number.
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
// nteams x G workgroups, at most one per CU (checked against the kernel's residency, never two team grids at
// once) and every poll is bounded -- a team that waits longer than ~10 s raises a fault word
// (host-mapped, reported by the next call on the handle) and all its members leave, so a lost
// workgroup can never hang the GPU.
//
// Teams are PERSISTENT: team t starts on tile t and then takes tiles from a queue until the batch is
// done, always in ITS OWN message slot (iteration 1 reads nothing, so a slot needs no clearing between
// tiles).  With one team per XCD the messages in flight are 7 or 8 slots, which the 256 MiB Infinity Cache keeps
// (tools/mall_probe.hip: the in-place sweeps run at 8.8 TB/s on a working set of <= 256 MiB against 5.1 ... 5.6 TB/s
// from HBM; the host plans the teams inside a budget of 240 MiB, team_fit() -- eight WHOLE slots of the n = 16384
// code, 8 x 32 MiB, fill the cache to the brim and every team is a fifth slower: seven teams then; with a quarter of
// a tile's rows on chip, "Rows in REGISTERS" below, eight fit).  The tile kernel cannot do that: one workgroup per
// tile means 768 tiles = 24 GiB in flight.
//
// Within a sweep a member has a share of the node chunks (every G-th chunk), and its waves take the chunks of
// that share one after the other from a counter in LDS (a wave's first chunk is its by right): the 8 waves of
// a member run at different speeds from sweep to sweep -- same work, different luck with the memory system --
// and a team barrier waits for the slowest wave of all.  (Dealing the chunks of the WHOLE team from one counter
// in global memory -- agent-scope atomics, or atomics without the scope bits that execute in the XCD's L2 --
// evens the finish out further but makes the sweeps themselves longer by more: profiles/r02_team_dealing_ab.txt.)
#pragma once
#include "bp_kernels.hpp"

#ifndef LDPC_TEAM_THREADS   // threads per member (experiments: 1024 = one 16-wave member per CU)
#define LDPC_TEAM_THREADS 512
#endif
#ifndef LDPC_TEAM_ROWS_WIDE   // 1 = the rows-in-LDS instantiation may use 256 VGPRs (it runs one workgroup per CU)
#define LDPC_TEAM_ROWS_WIDE 1
#endif
#ifndef LDPC_TEAM_TFORM   // 1 = the rows-on-chip instantiations of check degree <= 8 make the division of :147 in the variable sweep
#define LDPC_TEAM_TFORM 1
#endif
#ifndef LDPC_TEAM_TEST_DIRECT   // 1: the waves of a member OR their share of the convergence test into the team's word directly
#define LDPC_TEAM_TEST_DIRECT 1
#endif
#ifndef LDPC_TEAM_SLEEP   // s_sleep argument between two polls of the arrival counter (x64 cycles)
#define LDPC_TEAM_SLEEP 16
#endif

namespace ldpc {

// Rows in LDS (instantiations with LROWS: REGULAR graphs, every check exactly DC edges and every bit exactly DV --
// DC = 6 ... 10, DV = 3 ... 5, the codes of north_star's range: team_rows_degrees_ok()).  A member decodes the same share of
// the checks AND of the bits in every iteration, so an edge whose check and whose bit are both its own is never touched
// by anybody else: its message row can live in the member's LDS instead of the team's slot.  The host deals the bits
// to the members by the graph (team_rows_build(): a bit goes to a member that owns one of its checks -- 1 / DV of
// the edges become such) and gives the kernel the bit order, per check which of its edges are in LDS and from which
// LDS row on, per edge of a bit its LDS row or -1.  156 KiB of LDS hold R = 312 rows of 512 B per member: 15 % of a
// tile of the n = 16384 code -- that much less traffic through the XCD's port (full batch 1.013 -> 0.957 s).
// The two tables the sweeps read travel in the kernel's col_ptr / csc2csr arguments (const __restrict__: scalar loads;
// read through a pointer in a struct they became vector loads, one wait each, and the variable sweep took twice as long):
//   col_ptr  -> ctab [s][2]   per check: which of its edges are in LDS (bit k = edge k), the LDS row of the first of
//                             them (they follow each other)
//   csc2csr  -> vtab [n][VT]  per position p of the dealt bit order: CSR rows of its DV edges, their LDS rows or -1, the
//                             bit (| 1 << 31 when one of its edges is in LDS), padding to VT = 8 or 16 words
__host__ __device__ constexpr int team_vtab_words(int dv) { return 2 * dv + 1 <= 8 ? 8 : 16; }
// the degree pairs that have a rows-in-LDS instantiation (pick_team.hip)
__host__ __device__ constexpr bool team_rows_degrees_ok(int dc, int dv) { return dc >= 6 && dc <= 10 && dv >= 3 && dv <= 5; }

// One position record of vtab, read with as few scalar loads as its width allows (words 0 ... 7 in one, the rest in one more).
template <int DV>
struct TeamVRec {
    int pos[DV], lrow[DV], bit;
};
template <int DV>
__device__ __forceinline__ TeamVRec<DV> team_vrec_load(const int *__restrict__ vt)
{
    static_assert(2 * DV + 1 <= 16, "a position record is at most 16 words");
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v8i __attribute__((ext_vector_type(8)));
    constexpr int N = 2 * DV + 1;
    int w[16];
    const v8i A = *(const v8i *)vt;
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = A[k];
    if constexpr (N == 9) {
        w[8] = vt[8];
    } else if constexpr (N > 8 && N <= 10) {
        const v2i B = *(const v2i *)(vt + 8);
        w[8] = B[0]; w[9] = B[1];
    } else if constexpr (N > 10 && N <= 12) {
        const v4i B = *(const v4i *)(vt + 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[8 + k] = B[k];
    } else if constexpr (N > 12) {
        const v8i B = *(const v8i *)(vt + 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) w[8 + k] = B[k];
    }
    TeamVRec<DV> rc;
#pragma unroll
    for (int k = 0; k < DV; ++k) { rc.pos[k] = w[k]; rc.lrow[k] = w[DV + k]; }
    rc.bit = w[2 * DV];
    return rc;
}
// Rows in REGISTERS (instantiations with RR > 0, on top of the rows in LDS).  The LDS of a member holds 312 of the ~512
// rows that only it touches; the rest of them can live in the registers of ONE wave if that wave, and no other, updates
// both the row's check and its bit in every iteration.  So a part of every member's share of a sweep belongs to its
// waves BY RIGHT (the first static_c check chunks / static_v position chunks: chunk l of the share to wave l % W; the
// rest is dealt from the counter in LDS as before), and the host puts a bit whose owning check is such a wave's into
// a position that is the same wave's (team_rows_tables()).  A wave keeps up to kTeamRegRows rows in the TOP of its
// register file: row x in v[192 + 2x], v[193 + 2x], addressed with the wave-uniform row number through the VGPR index
// mode (s_set_gpr_idx_on + v_mov_b32).  The rows-on-chip instantiations run one 8-wave workgroup per CU (156 KB of
// LDS), i.e. two waves per SIMD: 256 registers a lane are theirs anyway, and the compiler needs about 130 of them and
// hands registers out from v0 upwards -- v192 ... v255 are free for the taking.  It only learns, from the clobber
// lists, that the kernel needs 256 registers; tests/test_abi_cpu.py checks in the built code object that nothing
// but these accessors touches v192 and up.  What did NOT work (all tried, all in the commit history of round 3):
//   * a C++ register array with a run-time index: arrays of more than 32 registers go to scratch memory; so does any
//     array whose address ends up in a select or phi, and "this register row or that memory row" is folded into
//     exactly that; kept in registers by force the arrays become SSA values of 32 registers that are copied at every
//     join (256 VGPRs and 888 bytes of spills);
//   * accumulator registers a0 ... a63 through the same index mode (gfx950 honours it for v_accvgpr_read / _write:
//     tools/agpr_index_probe.hip): once a kernel has accumulator registers at all, the register allocator parks its
//     own values in them between the asm statements -- 1,207 such moves, wrong results, the check sweep 20 us slower.
// With them 27 % of the rows of a (4,8)-regular tile are on chip instead of 15 %, eight slots fit the cache budget
// instead of seven, and -- the host gathers the on-chip rows into WHOLE CHECKS (team_rows_tables(): a bit goes to
// the owner of its first check) -- a quarter of the checks is updated without touching memory (check_update_regs, or
// check_update_exact on the LDS rows) and every bit has exactly its first edge on chip (bit_update_pair_first /
// bit_update_multi_first).  The general updates below (a pointer per edge; check_update_onchip, bit_update_onchip)
// remain for graphs and knob settings that spread the on-chip rows.
constexpr int kTeamRegRows = 32;
#define LDPC_TEAM_TOP_VGPRS                                                                                                      \
    "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207",     \
        "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", \
        "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", \
        "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
static_assert(192 + 2 * kTeamRegRows == 256, "the clobber list names the registers of kTeamRegRows rows");
// (every lane of the wave is active wherever these are used; row is wave-uniform.  s_nop: one wait state between the
// scalar write of M0 and a vector instruction that uses the index.  s_set_gpr_idx_on writes M0[7:0] and M0[15:12] --
// LLVM lists Defs = [M0, MODE] for it.  M0 is a RESERVED register to hipcc: naming it as a clobber only earns the
// warning that such clobbers "may not be preserved", so the accessors save M0 in a scalar register and put it back
// themselves -- whatever the compiler keeps in M0 survives them.  tests/test_abi_cpu.py checks in the code object that
// every s_set_gpr_idx_on is preceded by the save and every s_set_gpr_idx_off followed by the restore, with nothing
// but the s_nop and the two v_mov_b32 in between)
__device__ __forceinline__ double team_reg_get(const int row)
{
#ifdef LDPC_TEAM_FAKE_REGS   // (timing experiment: the code around the accessors without the accessors; results are wrong)
    return 0.5 + 1e-3 * row;
#endif
    unsigned int lo, hi, m0_keep;
    const int q = __builtin_amdgcn_readfirstlane(2 * row);
    asm volatile("s_mov_b32 %2, m0\n\ts_set_gpr_idx_on %3, 0x1\n\ts_nop 0\n\tv_mov_b32 %0, v192\n\tv_mov_b32 %1, v193\n\ts_set_gpr_idx_off\n\ts_mov_b32 m0, %2"
                 : "=v"(lo), "=v"(hi), "=&s"(m0_keep) : "s"(q) : LDPC_TEAM_TOP_VGPRS);
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ void team_reg_put(const int row, const double v)
{
#ifdef LDPC_TEAM_FAKE_REGS
    return;
#endif
    const unsigned int lo = (unsigned int)__double2loint(v), hi = (unsigned int)__double2hiint(v);
    const int q = __builtin_amdgcn_readfirstlane(2 * row);
    unsigned int m0_keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_set_gpr_idx_on %3, 0x8\n\ts_nop 0\n\tv_mov_b32 v192, %1\n\tv_mov_b32 v193, %2\n\ts_set_gpr_idx_off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_keep) : "v"(lo), "v"(hi), "s"(q) : LDPC_TEAM_TOP_VGPRS);
}
struct TeamRows {
    const int *lds_edge;        // [G][R]  CSR rows held by each member, -1 beyond its count (write-back before a hand-off)
    int R;                      // LDS rows per member
    const int *reg_edge;        // [G][W][regs]  CSR rows held by each wave in registers, -1 beyond its count
    int regs;                   // register rows per wave (0 ... kTeamRegRows)
    int static_c, static_v;     // chunks of a member's share of the check / variable sweep that its waves own by right
                                // (multiples of W, at least W, at most the smallest member's share)
    int first_c;                // the first first_c chunks of EVERY member's share of the check sweep are two checks with all their rows on
                                // chip (a multiple of W; the whole checks of a Gallager code's first block) -- see TeamParams::pre
    int flip;                   // bit 0 / 1: the upper half of a member's waves walks its chunks by right of the check /
                                // variable sweep from the last to the first (the chunks whose rows are all on chip come
                                // first in the dealt order: so one wave of every SIMD computes while the other one loads)
};

struct TeamParams {
    TeamRows rows;              // (LROWS instantiations only)
    int G;                      // workgroups per team
    int xcds;                   // XCDs that host teams (8; fewer = the blocks of the others leave at once, and so many fewer
                                // slots are in flight) -- not in scatter mode
    int nteams;                 // teams launched; team t works in message slot t (fresh tiles) and starts on tile t
    int pre;                    // chunks of on-chip checks (rows.first_c) that every wave updates BETWEEN telling the team that its variable
                                // sweep is through and waiting for the others (team_arrive / team_wait), when the next check sweep is
                                // certain to run: they need nobody else's rows, and the barrier's own round trips are idle time
    int pairs;                  // 1 = two nodes of the full degree are loaded together (twice the bytes in flight per wave)
    int dynamic;                // 1 = the waves of a member take the chunks of its share from a counter in LDS; 0 = every W-th
    // per team one control block of kTeamCtlWords words, zero at launch (see team_barrier); block number nteams
    // holds the tile queue (word 0: tiles handed out beyond the first nteams)
    unsigned int *ctl;
    u64 *mism;                  // [ntiles][mism_stride] mismatch words of the convergence tests, zero at launch
    int mism_stride;            // >= max_iters, a multiple of 32 words
    unsigned int *fault;        // host-mapped: set when a team barrier timed out
    int always_release;         // experiments: 1 = release at every barrier even when the team shares one XCD
    int scatter;                // 1 = deal a team's members over ALL XCDs (few tiles; also exercises the release path
                                // and the XCC check that selects it); gridDim.x == nteams * G
    // As a pass over a packed level (p.count_dev != nullptr): the number of syndromes is only known on the
    // device; the pass runs only for p.count_skip < *count_dev <= count_max (fewer: node kernel, more: the tile
    // kernel on the packed tiles), and its teams work IN the packed tiles (slot = tile).
    unsigned int count_max;
    int inject_fault;           // experiments build only (tests): 1 = raise the fault word and leave at once, as if a team
                                // barrier had timed out; 2 = the last member of team 0 stays away from the roll call
    unsigned int ticket;        // what a timed-out barrier writes into the fault word: the number of this call on its
                                // handle (low 31 bits, never 0), so that the report can name the first call that was hit;
                                // bit 31 set = the ROLL CALL failed: the call wrote no output at all
    unsigned int rollcall_ticks;   // how long (100 MHz ticks) the members of a team wait for each other at launch
    // Running ahead (fresh tiles): with the convergence test of an iteration under way a member goes straight on to the
    // NEXT iteration's check sweep -- the sweep does not depend on the test's outcome, only on the messages -- and the
    // barrier after that sweep serves the test too: two team barriers an iteration instead of three (C3 full-50:
    // 909 -> 882 ms).  The variable sweep after it must not overwrite the decision words the test's verdict still
    // captures from, hence a second set of them (errmask_alt: odd iterations).  A sweep ahead is wasted when the
    // verdict stops every lane, and it delays a straggler hand-off by an iteration (the messages are half an iteration
    // on); so a team only runs ahead while its tile is QUIET -- the previous verdict stopped no lane --, at least
    // ahead_min lanes are active, and the tile is at least two iterations short of where the team's PREVIOUS tile ended
    // (handed its stragglers on, or finished): tiles whose lanes have begun to converge, or are about to, take the
    // three-barrier iteration.  (Per 0.02, ~3 iterations a tile: running ahead on lane count alone cost 13 %, on
    // quietness alone 7 % -- the first verdict that is not quiet is usually the one that hands off.)
    u64 *errmask_alt;           // [ntiles][n] or nullptr (no running ahead: passes over packed levels)
    int ahead_min;              // active lanes from which on a quiet tile's team runs ahead (0 = never)
    int ahead_from;             // ... and the first iteration whose test may have company (1; 2: a verdict must have been quiet first)
    // LLRs (WANT_LLR instantiations).  0: log(1 / T) (:163, llr_of) per bit into p.llr[tile][bit][64], as the tile kernel
    // does (the passes over the packed levels).  4 (fresh tiles, the default) / 5 (decoders created with llr_exact): the
    // posterior odds T themselves -- their upper 32 bits (llr_hi32) / all 64 -- for every lane that is still active,
    // in every iteration, into p.llr's rows of the tile in POSITION-CHUNK layout: element (position, lane) at
    // ((position / 4) * 64 + lane) * 4 + position % 4, so that the four positions of a chunk are ONE 16-byte (32-byte)
    // store per lane; unpack_llr_kernel applies the position -> bit map of the dealt order, cuts / takes the logarithm
    // once per syndrome and bit, and transposes.  A lane that has stopped keeps what it stopped with.
    // (What this replaced, C3 full-50 / per 0.02 kernel time against 715 / 50 ms without LLRs -- DESIGN.md "LLRs": log(1 / T)
    // per bit and iteration 1000 / 64 ms; T as f64 per bit 862 / 61; 32-bit codes per bit 783 / 59; the teams' own scratch
    // rows + a copy-out per tile 745 / 57; this 752 / 55.)  6: experiments, no capture at all (LLRs undefined: what the
    // instantiation costs by itself -- nothing).
    int llr_raw;
};
constexpr unsigned int kTeamRollcallFailed = 0x80000000u;

// Team barrier number k (1, 2, ...).  Control block of a team: arrival counter at word 0; XCC mask, hand-off word
// and next-tile word at words 32, 33, 34; roll-call word at 36 (team_rollcall); one 128-byte line per member holding
// the number of the last barrier it was released from.
// The last arriver (it knows from the value its add returned) writes k into every member's line; the
// others poll THEIR OWN line.  (All members polling the one counter cost O(G^2) memory requests per
// barrier -- every add drops the line from every poller's L2 -- and with 8 ... 16 teams that storm took
// three times as long as the sweeps themselves.)
// one_xcd: every member of the team runs on the same XCD, so their stores meet in ONE L2 and the
// release (a write-back of that L2) is not needed; the acquire (this CU's L1) always is.
// Returns false when the team is broken (somebody timed out): the caller leaves the kernel.
constexpr int kTeamMaxMembers = 256;
constexpr int kTeamCtlMember = 64;                                  // first member line
constexpr int kTeamCtlWords = 64 + 32 * kTeamMaxMembers;            // per team
constexpr int kTeamCheckChunk = 2;                                  // checks per chunk of the check sweep

#ifndef LDPC_TEAM_BARRIER_COUNTER
#define LDPC_TEAM_BARRIER_COUNTER 0
#endif
__device__ __forceinline__ bool team_barrier(unsigned int *ctl, int G, int rank, unsigned int k, unsigned int *fault,
                                             unsigned int ticket, int *sh_ok, bool one_xcd, unsigned int *sh_deal)
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
        // (Experiment, -DLDPC_TEAM_BARRIER_COUNTER=1: a team inside one XCD -- one L2 -- waits on the arrival counter
        // itself, so that the last arrival is seen one round trip earlier than through a flag the last member writes for
        // everybody after its own fetch_add has come back.  Measured SLOWER, round 4, alternating builds on one box: C3
        // full-50 720.9 / 721.0 against 713.4 / 714.3 ms, (3,6) n = 16380 514.9 against 505.3 ms -- 31 pollers on the line
        // hold up the 32 adds more than the round trip saves.  Off.)
        const bool on_counter = LDPC_TEAM_BARRIER_COUNTER && one_xcd;
        if (prev + 1u == k * (unsigned)G) {
            if (!on_counter)
                for (int m = 0; m < G; ++m)
                    __hip_atomic_store(ctl + kTeamCtlMember + 32 * m, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned int *mine = on_counter ? ctl : ctl + kTeamCtlMember + 32 * rank;
            const unsigned int want = on_counter ? k * (unsigned)G : k;
            const u64 t0 = wall_clock64();
            unsigned int polls = 0;
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
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
        sh_deal[0] = 0u;   // no wave of this workgroup is inside a sweep here: the dealers start afresh
        sh_deal[1] = 0u;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// The two halves of team_barrier for a caller with work of its own between them (TeamParams::pre): what a member does after
// team_arrive must touch nothing that another member reads or writes.
__device__ __forceinline__ void team_arrive(unsigned int *ctl, int G, unsigned int k, bool one_xcd)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its own stores
    __syncthreads();
    if (threadIdx.x == 0) {
        if (!one_xcd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned int prev = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == k * (unsigned)G)
            for (int m = 0; m < G; ++m)
                __hip_atomic_store(ctl + kTeamCtlMember + 32 * m, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool team_wait(unsigned int *ctl, int rank, unsigned int k, unsigned int *fault, unsigned int ticket,
                                          int *sh_ok, unsigned int *sh_deal)
{
    if (threadIdx.x == 0) {
        int ok = 1;
        unsigned int *mine = ctl + kTeamCtlMember + 32 * rank;   // (the last arriver finds its own flag set)
        const u64 t0 = wall_clock64();
        unsigned int polls = 0;
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k) {
            __builtin_amdgcn_s_sleep(LDPC_TEAM_SLEEP);
            if ((++polls & 255u) == 0u) {                        // (as in team_barrier)
                if (__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { ok = 0; break; }
                if (wall_clock64() - t0 > 1000000000ull) {
                    __hip_atomic_store(fault, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        *sh_ok = ok;
        sh_deal[0] = 0u;
        sh_deal[1] = 0u;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// Launch-time residency handshake.  A team barrier only works when all G members are on the machine at once.  The
// host sizes the grid so that they fit and never runs two team grids of one process together, but it cannot see what
// OTHER processes have put on the GPU: a member may be waiting for a CU that a foreign kernel holds.  So the first
// thing a team does is a roll call with a SHORT bound (tp.rollcall_ticks: milliseconds): every member adds itself to
// the team's roll-call word and waits until all G are counted.  A member that runs out of patience marks the word
// DEAD -- by compare-and-swap, which can only succeed while the count is still short, so "dead" and "complete" exclude
// each other -- raises the fault word (ticket | kTeamRollcallFailed) and leaves; every other member sees the mark (a
// late arriver in the value its add returns) and leaves too.  Nothing has been read or written at that point: the
// host decodes the batch with the tile kernel, which needs no co-residency, instead of finding out from a barrier
// that times out after 10 s.  Co-tenancy with other processes' kernels therefore costs throughput, not correctness.
__device__ __forceinline__ bool team_rollcall(unsigned int *word, int G, unsigned int *fault, unsigned int ticket,
                                              unsigned int ticks, int *sh_ok)
{
    constexpr unsigned int DEAD = 0x80000000u;
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned int prev = __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev & DEAD) {
            ok = 0;                                            // somebody gave up before this member got a CU
        } else if (prev + 1u != (unsigned)G) {
            const u64 t0 = wall_clock64();
            for (unsigned int polls = 1;; ++polls) {
                unsigned int c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c & DEAD) { ok = 0; break; }
                if (c >= (unsigned)G) break;                   // everybody is here
                __builtin_amdgcn_s_sleep(8);
                if ((polls & 31u) == 0u && wall_clock64() - t0 > (u64)ticks) {
                    bool dead = false;
                    while (c < (unsigned)G && !dead)           // (c is refreshed by a failed exchange)
                        dead = __hip_atomic_compare_exchange_strong(word, &c, c | DEAD, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                    __HIP_MEMORY_SCOPE_AGENT);
                    if (dead) __hip_atomic_store(fault, ticket | kTeamRollcallFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (dead || (c & DEAD)) { ok = 0; break; }
                }
            }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// The next chunk of this member's share for the calling wave (beyond the W the waves own by right).
__device__ __forceinline__ int team_deal(unsigned int *counter, int lane)
{
    unsigned int t = 0;
    if (lane == 0) t = atomicAdd(counter, 1u);   // LDS
    return __builtin_amdgcn_readfirstlane((int)t);
}

// Check / bit updates with some rows in LDS: generic pointers, the hardware routes each access.  Same arithmetic.
template <int D, bool FIRST, bool TF = false>
__device__ __forceinline__ void check_update_mixed(double *M, double *L, unsigned int mask, double sigma, double r)
{
    double *ptr[D];
    int nl = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const bool in_lds = (mask >> k) & 1u;
        ptr[k] = in_lds ? L + (size_t)nl * kTile : M + (size_t)k * kTile;
        nl += in_lds ? 1 : 0;
    }
    double a[D], out[D];
    if (FIRST) {
        const double a0 = 2.0 / (1.0 + r) - 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = a0;
    } else {
        double m[D];
#pragma unroll
        for (int k = 0; k < D; ++k) m[k] = *ptr[k];
        check_factors<D>(m, a);
    }
    check_compute_exact<D, TF>(a, sigma, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) *ptr[k] = out[k];
}

template <int D, bool TF = false>
__device__ __forceinline__ double bit_update_mixed(double *Mt, double *L, const int (&pos)[D], const int (&lrow)[D], double r)
{
    double *ptr[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const bool in_lds = lrow[k] >= 0;
        ptr[k] = (in_lds ? L : Mt) + (size_t)(in_lds ? lrow[k] : pos[k]) * kTile;
    }
    double c[D], out[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = *ptr[k];
    if (TF) check_to_odds<D>(c, c);   // :147
    const double F = bit_compute_exact<D>(c, r, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) *ptr[k] = out[k];
    return F;
}

// ... and with some rows in the wave's registers (rget(row) reads register row `row`, rput(on, row, v) writes it when
// `on`; row numbers are wave-uniform): rmask = the check's edges that live there, from register row rbase on.  A
// row that lives in registers is neither loaded from nor stored to its place in the slot: its pointer is bent to a
// dummy row of the member's LDS (Ld), so that the loads and stores of the check's other rows need no branches around
// them and are all in flight together -- with "if (in memory) load" per edge the compiler serialises them, and a
// check with ONE row in registers (the usual case: a member's on-chip edges are spread over its checks) cost 5 us.
// Same arithmetic.
template <int D, bool FIRST, bool TF, class RGet, class RPut>
__device__ __forceinline__ void check_update_onchip(double *M, double *L, double *Ld, unsigned int mask, unsigned int rmask, int rbase,
                                                    double sigma, double r, RGet &&rget, RPut &&rput)
{
    double *ptr[D];
    int nl = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const bool in_lds = (mask >> k) & 1u, in_reg = (rmask >> k) & 1u;
        ptr[k] = in_reg ? Ld : (in_lds ? L + (size_t)nl * kTile : M + (size_t)k * kTile);
        nl += in_lds ? 1 : 0;
    }
    double a[D], out[D];
    if (FIRST) {
        const double a0 = 2.0 / (1.0 + r) - 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = a0;
    } else {
        double m[D];
#pragma unroll
        for (int k = 0; k < D; ++k) m[k] = *ptr[k];
#pragma unroll
        for (int k = 0; k < D; ++k)
            if ((rmask >> k) & 1u) m[k] = rget(rbase + __builtin_popcount(rmask & ((1u << k) - 1u)));
        check_factors<D>(m, a);
    }
    check_compute_exact<D, TF>(a, sigma, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) *ptr[k] = out[k];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) rput((rmask >> k) & 1u, rbase + __builtin_popcount(rmask & ((1u << k) - 1u)), out[k]);
}

// loc[k]: >= 0 that LDS row, -1 the slot (row pos[k]), <= -2 register row -2 - loc[k] of this wave
template <int D, bool TF, class RGet, class RPut>
__device__ __forceinline__ double bit_update_onchip(double *Mt, double *L, double *Ld, const int (&pos)[D], const int (&loc)[D], double r,
                                                    RGet &&rget, RPut &&rput)
{
    double *ptr[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const bool in_lds = loc[k] >= 0;
        ptr[k] = loc[k] <= -2 ? Ld : (in_lds ? L : Mt) + (size_t)(in_lds ? loc[k] : pos[k]) * kTile;
    }
    double c[D], out[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = *ptr[k];
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (loc[k] <= -2) c[k] = rget(-2 - loc[k]);
    if (TF) check_to_odds<D>(c, c);   // :147
    const double F = bit_compute_exact<D>(c, r, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) *ptr[k] = out[k];
#pragma unroll
    for (int k = D - 1; k >= 0; --k) rput(loc[k] <= -2, -2 - loc[k], out[k]);
    return F;
}

// A check whose rows ALL live in this wave's registers (rows rbase ... rbase + D - 1): no memory at all.  This is
// the usual kind when the host gathers the on-chip rows into whole checks (team_rows_tables(), "concentrate").
template <int D, bool FIRST, bool TF, class RGet, class RPut>
__device__ __forceinline__ void check_update_regs(int rbase, double sigma, double r, RGet &&rget, RPut &&rput)
{
    double a[D], out[D];
    if (FIRST) {
        const double a0 = 2.0 / (1.0 + r) - 1.0;
#pragma unroll
        for (int k = 0; k < D; ++k) a[k] = a0;
    } else {
        double m[D];
#pragma unroll
        for (int k = 0; k < D; ++k) m[k] = rget(rbase + k);
        check_factors<D>(m, a);
    }
    check_compute_exact<D, TF>(a, sigma, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) rput(true, rbase + k, out[k]);
}

// Two bits whose edges 1 ... D-1 are rows of the slot and whose FIRST edge is anywhere (loc >= 0: that LDS row, -1:
// the slot, <= -2: register row -2 - loc of this wave) -- what the variable sweep meets when the on-chip rows are whole
// checks of the first block: every bit has exactly its first edge there.  All 2 (D - 1) slot rows are in flight
// together (scalar row numbers: no pointer per edge as in bit_update_onchip); the first edges come by a wave-uniform
// branch each.  Same arithmetic per bit.
template <int D, bool TF, class RGet, class RPut>
__device__ __forceinline__ void bit_update_pair_first(double *Mt, double *L, const int (&pos0)[D], const int loc0, const int (&pos1)[D],
                                                      const int loc1, double r, RGet &&rget, RPut &&rput, double &T0, double &T1)
{
    double c0[D], c1[D], o0[D], o1[D];
    size_t at0[D], at1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { at0[k] = (size_t)pos0[k] * kTile; at1[k] = (size_t)pos1[k] * kTile; }
#pragma unroll
    for (int k = 1; k < D; ++k) c0[k] = ldm(Mt + at0[k]);
#pragma unroll
    for (int k = 1; k < D; ++k) c1[k] = ldm(Mt + at1[k]);
    if (loc0 == -1) c0[0] = ldm(Mt + at0[0]);
    else if (loc0 >= 0) c0[0] = L[(size_t)loc0 * kTile];
    else c0[0] = rget(-2 - loc0);
    if (loc1 == -1) c1[0] = ldm(Mt + at1[0]);
    else if (loc1 >= 0) c1[0] = L[(size_t)loc1 * kTile];
    else c1[0] = rget(-2 - loc1);
    loads_first();
    if (TF) { check_to_odds<D>(c0, c0); check_to_odds<D>(c1, c1); }   // :147
    T0 = bit_compute_exact<D>(c0, r, o0);
#pragma unroll
    for (int k = D - 1; k >= 1; --k) stm(Mt + at0[k], o0[k]);
    if (loc0 == -1) stm(Mt + at0[0], o0[0]);
    else if (loc0 >= 0) L[(size_t)loc0 * kTile] = o0[0];
    else rput(true, -2 - loc0, o0[0]);
    T1 = bit_compute_exact<D>(c1, r, o1);
#pragma unroll
    for (int k = D - 1; k >= 1; --k) stm(Mt + at1[k], o1[k]);
    if (loc1 == -1) stm(Mt + at1[0], o1[0]);
    else if (loc1 >= 0) L[(size_t)loc1 * kTile] = o1[0];
    else rput(true, -2 - loc1, o1[0]);
}

// ... and NB bits at once (the four positions of a chunk): all NB (D - 1) slot rows in flight before anything is computed.
// The variable sweep waits for scattered rows; a wave that asks for 6 of them at a time leaves the fabric idle.
// KIND: where the first edges of ALL NB bits are -- 1: in LDS, 2: in this wave's registers, 0: anywhere (a wave-uniform
// branch per bit and side).  The host fills a wave's register rows with the bits of whole position chunks
// (team_rows_tables()), so nearly every chunk is of one kind and takes straight-line code.
template <int D, int NB, bool TF, int KIND, class RGet, class RPut>
__device__ __forceinline__ void bit_update_multi_first(double *Mt, double *L, const int (&pos)[NB][D], const int (&loc)[NB], double r,
                                                       RGet &&rget, RPut &&rput, double (&T)[NB])
{
    double c[NB][D];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int k = 1; k < D; ++k) c[b][k] = ldm(Mt + (size_t)pos[b][k] * kTile);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (KIND == 1) c[b][0] = L[(size_t)loc[b] * kTile];
        else if (KIND == 2) c[b][0] = rget(-2 - loc[b]);
        else if (loc[b] == -1) c[b][0] = ldm(Mt + (size_t)pos[b][0] * kTile);
        else if (loc[b] >= 0) c[b][0] = L[(size_t)loc[b] * kTile];
        else c[b][0] = rget(-2 - loc[b]);
    }
    loads_first();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (TF) check_to_odds<D>(c[b], c[b]);   // :147
        double o[D];
        T[b] = bit_compute_exact<D>(c[b], r, o);
#pragma unroll
        for (int k = D - 1; k >= 1; --k) stm(Mt + (size_t)pos[b][k] * kTile, o[k]);
        if (KIND == 1) L[(size_t)loc[b] * kTile] = o[0];
        else if (KIND == 2) rput(true, -2 - loc[b], o[0]);
        else if (loc[b] == -1) stm(Mt + (size_t)pos[b][0] * kTile, o[0]);
        else if (loc[b] >= 0) L[(size_t)loc[b] * kTile] = o[0];
        else rput(true, -2 - loc[b], o[0]);
    }
}

// A bit in the first iteration of a fresh tile (LDPC_TEAM_FUSE_FIRST): its D incoming messages are table[sign of the check][place
// of the edge in its check] (first[0 ... DC-1]: syndrome bit 0, first[DC ... 2 DC-1]: syndrome bit 1), CSR row q = DC * check + place.
// loc[k] as in bit_update_onchip.  Same arithmetic as a bit whose rows the first check sweep had filled.
template <int D, int DC, bool TF, class RPut>
__device__ __forceinline__ double bit_update_first(double *Mt, double *L, const double *first, const u64 *__restrict__ syn, int lane,
                                                   const int (&pos)[D], const int (&loc)[D], double r, RPut &&rput)
{
    double c[D], out[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int i = pos[k] / DC, place = pos[k] - i * DC;
        const bool odd = (syn[i] >> lane) & 1ull;
        c[k] = first[(odd ? DC : 0) + place];
    }
    if (TF) check_to_odds<D>(c, c);   // :147
    const double F = bit_compute_exact<D>(c, r, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) {
        if (loc[k] == -1) stm(Mt + (size_t)pos[k] * kTile, out[k]);
        else if (loc[k] >= 0) L[(size_t)loc[k] * kTile] = out[k];
        else rput(true, -2 - loc[k], out[k]);
    }
    return F;
}

// IRREGULAR graphs with rows in LDS (instantiations with IRR; round 4).  The host packs WHOLE checks into the LDS of their
// owners (team_irr_tables(): a check qualifies when every one of its bits can be given to the check's owner -- a set
// packing over the checks -- and the member has LDS rows and positions left) and relabels the bits by position, as the
// regular tables do.  The kernel reads its graph through the same four arguments, in these forms:
//   row_ptr  -> ctab2 [s + 1][2]   {first CSR row of the check, its first LDS row or -1 (all its rows are consecutive there)}
//   col_ptr  -> ptab  [n + 1][2]   {first entry of the position's edge list, the bit | 1 << 31 when one of its edges is in LDS}
//   csc2csr  -> ploc  [nnz]        per edge of a position (checks ascending): its CSR row, or -1 - (LDS row) of the member
//   edge_bit                       unchanged (the convergence test)
// A check in LDS takes the ordinary update on a generic pointer (the hardware routes LDS and global accesses alike); a
// bit with an edge in LDS takes bit_update_flat (a pointer per edge).  Nodes wider than the register buckets never
// get rows in LDS (the host sees to it), so the O(deg^2) paths stay on the slot.
#ifndef LDPC_TEAM_TEST_SPREAD
#define LDPC_TEAM_TEST_SPREAD 1
#endif
#ifndef LDPC_TEAM_FAULT_BLOCKS   // the two early exits of the fault injection (tests): compiled into the experiments build
#ifdef LDPC_EXPERIMENTS
#define LDPC_TEAM_FAULT_BLOCKS 1
#else
#define LDPC_TEAM_FAULT_BLOCKS 0
#endif
#endif
// Where the compiler puts the basic blocks of the sweeps matters by 2 %: the headline instantiation ran 725 ms as it was and
// 709-713 ms under ANY of four small perturbations of the source (the experiments build's two early exits, these hints, both,
// straight-line chunk kinds) -- round 4, alternating builds on one box, profiles/r04_tform_ab.txt.  The hints say what is
// true -- nearly every chunk of either sweep is the usual one -- and measured best: C3 full-50 724.9 -> 709.0 ms, per 0.02
// 50.2 -> 49.0 ms; (3,6) n = 16380 and (4,10) n = 16380 unchanged (+0.2 %).
#ifndef LDPC_TEAM_LIKELY
#define LDPC_TEAM_LIKELY 1
#endif
// The FIRST iteration of a fresh tile without its check sweep (rows-on-chip kernels: regular graphs).  Every bit -> check
// message is still r (:129), so what the first check sweep would store in row k of a check depends on the check's syndrome bit
// and on k alone: 2 DC values, made once per workgroup by the same instruction sequence (check_compute_exact).  The first
// variable sweep takes them from that table in LDS instead of loading rows that a sweep before it would have had to store:
// one sweep of stores, one of loads and a team barrier less per tile -- a seventh of the traffic of a tile that converges in
// three to four iterations.
#ifndef LDPC_TEAM_FUSE_FIRST
#define LDPC_TEAM_FUSE_FIRST 1
#endif
#ifndef LDPC_TEAM_FUSE_DC_MAX
#define LDPC_TEAM_FUSE_DC_MAX 10
#endif
#ifndef LDPC_TEAM_LIKELY_DC_MAX
#define LDPC_TEAM_LIKELY_DC_MAX 10
#endif
// (check degree 10 lost by them at first -- (5,10) n = 16000 491.2 against 487.2 ms, (4,10) 353.9 against 353.1 -- and has them
//  since the fused first iteration: with it and without them (4,10) ran 357.6 ms, with both 346.6, with neither 348.1;
//  kTeamHints in the kernel)
#define LDPC_HOT(x) (kTeamHints ? __builtin_expect(!!(x), 1) : !!(x))
// The four first edges of a position chunk all in LDS / all in this wave's registers: straight-line code for the chunk
// (bit_update_multi_first<KIND>) instead of a wave-uniform branch per bit and side.  -1: for bit degree 3 only, where it
// measured -1 % ((3,6) n = 16380: 490.4 against 494.6 ms); C3 +0.4 %, (4,10) +1 % slower with it.
#ifndef LDPC_TEAM_CHUNK_KINDS
#define LDPC_TEAM_CHUNK_KINDS -1
#endif

template <int D>
__device__ __forceinline__ double bit_update_exact_flat(double *Mt, double *L, const int *__restrict__ loc, double r)
{
    double *ptr[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const int q = loc[k];
        ptr[k] = q >= 0 ? Mt + (size_t)q * kTile : L + (size_t)(-1 - q) * kTile;
    }
    double c[D], out[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = *ptr[k];
    const double F = bit_compute_exact<D>(c, r, out);
#pragma unroll
    for (int k = D - 1; k >= 0; --k) *ptr[k] = out[k];
    return F;
}
template <int LO, int HI>
__device__ __forceinline__ double bit_dispatch_flat(double *Mt, double *L, const int *__restrict__ loc, int deg, double r)
{
    if constexpr (LO == HI) {
        return bit_update_exact_flat<LO>(Mt, L, loc, r);
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (deg <= MID) return bit_dispatch_flat<LO, MID>(Mt, L, loc, deg, r);
        return bit_dispatch_flat<MID + 1, HI>(Mt, L, loc, deg, r);
    }
}
template <int DV>
__device__ __forceinline__ double bit_update_flat(double *Mt, double *L, const int *__restrict__ loc, int deg, double r)
{
    if (deg == 0) return r;
    return bit_dispatch_flat<1, DV>(Mt, L, loc, deg <= DV ? deg : DV, r);   // (deg <= DV: the host's promise for such bits)
}

// Register budget: teams run one or two workgroups per CU (team_geometry() on the host), so the narrow-degree
// instantiation may have 128 VGPRs instead of the tile kernel's 80 (three workgroups per CU) -- under 80 it spilled.
template <int DC, int DV, int THREADS, bool LROWS = false>
constexpr int team_min_waves_per_simd()
{
    if (LROWS && LDPC_TEAM_ROWS_WIDE) return THREADS / 256;   // 156 KB of LDS: one workgroup per CU anyway
    return min_waves_per_simd<DC, DV, THREADS>() < 4 ? min_waves_per_simd<DC, DV, THREADS>() : 4;
}

// RESUMED: the pass over a packed level (the messages are in the packed tiles, every lane has iterations behind it)
// is an instantiation of its own -- it shows under its own name in a profile, and the fresh pass loses the tests.
template <int DC, int DV, bool WANT_LLR, int THREADS, bool RESUMED, bool LROWS = false, int RR = 0, bool IRR = false>
__global__ void
__launch_bounds__(THREADS, (team_min_waves_per_simd<DC, DV, THREADS, LROWS || IRR>()))
bp_team_kernel(BPParams p, TeamParams tp, const int *__restrict__ row_ptr, const int *__restrict__ edge_bit,
               const int *__restrict__ col_ptr, const int *__restrict__ csc2csr,
               const u64 *__restrict__ synmask, const u64 *__restrict__ nevermask)
{
    static_assert(!(LROWS && RESUMED), "packed tiles are decoded where they lie, all rows in global memory");
    static_assert(RR == 0 || (LROWS && RR == kTeamRegRows), "rows in registers come on top of the rows in LDS");
    static_assert(!IRR || (!LROWS && !RESUMED && RR == 0), "irregular graphs: whole checks in LDS, fresh tiles");
    constexpr int RS = IRR ? 2 : 1;   // stride of row_ptr / col_ptr entries (IRR: pairs, see above)
    constexpr bool kTeamHints = (LDPC_TEAM_LIKELY != 0) && DC <= LDPC_TEAM_LIKELY_DC_MAX;   // (LDPC_HOT, above)
    // Where the division (1 - t) / (1 + t) of :147 is made (check_finish_exact): the rows-on-chip instantiations leave
    // it to the variable sweep.  A check costs 16 fp64 divisions (two per edge) and the check sweep of the persistent
    // teams is bound by them, not by the memory side; the variable sweep has no division at all and waits for its
    // scattered rows.  One division per edge in either sweep: the same operations on the same operands, the same
    // bits -- and at every iteration boundary (hand-off, write-back) the rows are bit -> check messages as ever.
    // Measured on the final kernels of round 4, alternating builds on one box (profiles/r04_tform_ab.txt): C3 full-50
    // 721.4 -> 713.9 ms, (3,6) n = 16380 521.6 -> 515.1 ms, (5,10) n = 16000 491.0 -> 493.1 ms -- hence check degree <= 8.
    // (Round 3, before the on-chip rows were whole checks: 873 -> 876 ms, and the form stayed off.)
    constexpr bool TF = LROWS && (LDPC_TEAM_TFORM != 0) && DC <= 8;
    // rows in registers (RR > 0): v[192 + 2 * row], v[193 + 2 * row] of this wave (team_reg_get / team_reg_put)
    auto rget = [&](const int row) -> double {
        if constexpr (RR > 0) return team_reg_get(row);
        else return 0.0;
    };
    auto rput = [&](const bool on, const int row, const double v) {   // on: this edge lives in a register row (else nothing happens)
        if constexpr (RR > 0) {
            if (on) team_reg_put(row, v);
        }
    };
    extern __shared__ double lds_rows[];   // LROWS: [tp.rows.R][64]
    double *const Lr = lds_rows + (threadIdx.x & 63);
    double *const Ldummy = Lr + (size_t)tp.rows.R * kTile;   // (RR > 0: one row past the member's LDS rows, see check_update_onchip)
    __shared__ int sh_ok;
    __shared__ unsigned int sh_deal[2];   // chunks of this member's share dealt so far beyond the waves' first: check sweep, variable sweep
#if !LDPC_TEAM_TEST_DIRECT
    __shared__ u64 sh_mism[THREADS / 64];
#endif
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int W = THREADS / 64;
    const int s = p.s, n = p.n;
    const double r = p.r;
    constexpr bool kFuseFirst = LROWS && (LDPC_TEAM_FUSE_FIRST != 0) && DC <= LDPC_TEAM_FUSE_DC_MAX;   // (LROWS: fresh tiles only, every check of degree DC)
    __shared__ double sh_first[kFuseFirst ? 2 * DC : 1];
    if constexpr (kFuseFirst) {
        if (threadIdx.x < 2) {   // what the first check sweep stores in the rows of a check: syndrome bit 0 / 1 (check_update_exact<DC, true, TF>)
            const double a0 = 2.0 / (1.0 + p.r) - 1.0;
            double a[DC], out[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k) a[k] = a0;
            check_compute_exact<DC, TF>(a, threadIdx.x ? -1.0 : 1.0, out);
#pragma unroll
            for (int k = 0; k < DC; ++k) sh_first[threadIdx.x * DC + k] = out[k];
        }
    }
    if (threadIdx.x == 0) { sh_deal[0] = 0u; sh_deal[1] = 0u; }
    __syncthreads();
#if LDPC_TEAM_FAULT_BLOCKS
    if (tp.inject_fault == 1) {
        if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(tp.fault, tp.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
#endif
    const int G = tp.G, nteams = tp.nteams;
    int ntiles = p.ntiles;
    long long batch = p.batch;
    if (p.count_dev) {                                         // a pass over a packed level: sized on the device
        batch = (long long)*p.count_dev;
        if (batch <= (long long)p.count_skip || batch > (long long)tp.count_max) return;
        ntiles = (int)((batch + kTile - 1) / kTile);
    }
    // Workgroups are dealt round-robin over the 8 XCDs (observed, not promised): blocks b and b + 8 share
    // one.  Teams are formed among the blocks of one residue class, so that normally a team sits on ONE
    // XCD; the members check it (xccs) and fall back to full release / acquire barriers if it is not so.
    const int bq = (int)(blockIdx.x >> 3), xslot = (int)(blockIdx.x & 7u);   // gridDim.x == 8 * G * ceil(nteams / 8)
    if (!tp.scatter && xslot >= tp.xcds) return;
    const int team = tp.scatter ? (int)(blockIdx.x / (unsigned)G) : (bq / G) * tp.xcds + xslot;
    const int rank = tp.scatter ? (int)(blockIdx.x % (unsigned)G) : bq % G;
    if (team >= nteams || team >= ntiles) return;              // whole teams only: nobody waits for these
    const int gw = rank * W + w, GW = G * W;                   // this wave among the team's waves
    unsigned int *const ctr = tp.ctl + (size_t)team * kTeamCtlWords;
    unsigned int *const xccs = ctr + 32;
    unsigned int *const tile_queue = tp.ctl + (size_t)nteams * kTeamCtlWords;
    unsigned int epoch = 0;                                    // barriers passed
    __shared__ int sh_one_xcd;
#if LDPC_TEAM_FAULT_BLOCKS
    if (tp.inject_fault == 2 && team == 0 && rank == G - 1) return;   // (tests) a member that never gets its CU
#endif
    if (!team_rollcall(ctr + 36, G, tp.fault, tp.ticket, tp.rollcall_ticks, &sh_ok)) return;
    if (threadIdx.x == 0) {
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_fetch_or(xccs, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool one_xcd = false, placed = false;                      // known after the first (full) barrier
    constexpr bool resumed = RESUMED;                          // (the host sets p.resumed alike)
    const BPCold *const cd = p.cold;
    u64 tk_check = 0, tk_var = 0, tk_rest = 0;   // this wave's own sweep time / everything else (waiting included)
    int horizon = p.max_iters;                   // the iteration at which this team's previous tile ended (running ahead: TeamParams)
    long long tot_iters = 0;                     // (rank 0, wave 0) iterations of the syndromes this team finished

    for (int tile = team; tile < ntiles;) {
    // fresh tiles are decoded in the team's own slot, packed tiles of a level where they lie
    double *const Mt = p.msg + (size_t)(resumed ? tile : team) * (size_t)p.slot_stride + lane;
    u64 *const mw = tp.mism + (size_t)tile * (size_t)tp.mism_stride;
    const u64 *syn = synmask + (size_t)tile * s;
    u64 *em = p.errmask + (size_t)tile * n;   // decision words of the iteration in hand (see em_even / em_odd)
    u64 *fin = p.finmask + (size_t)tile * n;
    const long long b0 = (long long)tile * kTile;
    const long long left = batch - b0;
    u64 deferred = 0;
    const u64 valid = left >= kTile ? ~0ull : ((1ull << left) - 1ull);
    const u64 never = nevermask[tile];
    u64 active = valid;                                        // identical in every member: same inputs, same words
    // a pass over a packed level: every lane has it0 iterations behind it and its messages in the packed tile
    const int it0 = (resumed && ((valid >> lane) & 1ull)) ? cd->it0[b0 + lane] : 0;
    int my_iters = 0, my_conv = 0, it = 0;

    // running ahead (TeamParams::errmask_alt): decision words of odd iterations live in the second set
    const bool can_run_ahead = !resumed && tp.errmask_alt != nullptr && tp.ahead_min > 0;
    u64 *const em_even = em, *const em_odd = can_run_ahead ? tp.errmask_alt + (size_t)tile * n : em;
    bool have_check = false;                                   // this iteration's check sweep was done ahead, with the last test
    bool quiet = true;                                         // the last verdict stopped no lane (running ahead: TeamParams)

    // ---- check-node sweep  (:135-150) in chunks of kTeamCheckChunk checks
    auto check_sweep = [&](const bool first, const int skip = 0) {   // skip: this wave's first `skip` chunks were updated already (check_pre)
        const int nch = (s + kTeamCheckChunk - 1) / kTeamCheckChunk;
        auto chunk = [&](int c) {
            const int i1 = min(s, (c + 1) * kTeamCheckChunk);
            if constexpr (LROWS) {
                // every check has the full degree (the host instantiates this only for such graphs); col_ptr is ctab
                // [s][4] = {edges in LDS, first LDS row, edges in registers, first register row}.
                // (Pairs of checks / bits with rows in LDS loaded together measured slower, not faster.)
                typedef int v4i __attribute__((ext_vector_type(4)));
                typedef int v8i __attribute__((ext_vector_type(8)));
                constexpr int FULL = (1 << DC) - 1;
                auto one = [&](double *const M, const v4i ct, const double sg) {
                    if (RR > 0 && ct.z == FULL) {            // the whole check in this wave's registers
                        if (first) check_update_regs<DC, true, TF>(ct.w, sg, r, rget, rput);
                        else check_update_regs<DC, false, TF>(ct.w, sg, r, rget, rput);
                    } else if (ct.x == FULL) {               // the whole check in LDS: consecutive rows from ct.y on
                        if (first) check_update_exact<DC, true, TF>(Lr + (size_t)ct.y * kTile, sg, r);
                        else check_update_exact<DC, false, TF>(Lr + (size_t)ct.y * kTile, sg, r);
                    } else if (RR > 0 && ct.z != 0) {
                        if (first) check_update_onchip<DC, true, TF>(M, Lr + (size_t)ct.y * kTile, Ldummy, (unsigned int)ct.x, (unsigned int)ct.z, ct.w, sg, r, rget, rput);
                        else check_update_onchip<DC, false, TF>(M, Lr + (size_t)ct.y * kTile, Ldummy, (unsigned int)ct.x, (unsigned int)ct.z, ct.w, sg, r, rget, rput);
                    } else if (first) {
                        check_update_mixed<DC, true, TF>(M, Lr + (size_t)ct.y * kTile, (unsigned int)ct.x, sg, r);
                    } else if (ct.x == 0) {
                        check_update_exact<DC, false, TF>(M, sg, r);
                    } else {
                        check_update_mixed<DC, false, TF>(M, Lr + (size_t)ct.y * kTile, (unsigned int)ct.x, sg, r);
                    }
                };
                const int i = c * kTeamCheckChunk;
                if (LDPC_HOT(kTeamCheckChunk == 2 && i + 2 == i1)) {
                    const v8i ct = *(const v8i *)(col_ptr + 4 * i);   // both checks' table rows in one scalar load
                    const u64 s0 = syn[i], s1 = syn[i + 1];
                    const double sg0 = ((s0 >> lane) & 1ull) ? -1.0 : 1.0, sg1 = ((s1 >> lane) & 1ull) ? -1.0 : 1.0;
                    double *const M0 = Mt + (size_t)i * DC * kTile, *const M1 = M0 + (size_t)DC * kTile;
                    if ((ct.s0 | ct.s2 | ct.s4 | ct.s6) == 0 && !first && tp.pairs) {   // (a hint here: no difference)
                        check_update_pair<DC, TF>(M0, M1, sg0, sg1);
                    } else {
                        one(M0, v4i{ct.s0, ct.s1, ct.s2, ct.s3}, sg0);
                        one(M1, v4i{ct.s4, ct.s5, ct.s6, ct.s7}, sg1);
                    }
                    return;
                }
                for (int ii = i; ii < i1; ++ii) {
                    const v4i ct = *(const v4i *)(col_ptr + 4 * ii);
                    one(Mt + (size_t)ii * DC * kTile, ct, ((syn[ii] >> lane) & 1ull) ? -1.0 : 1.0);
                }
                return;
            }
            if constexpr (IRR) {
                for (int i = c * kTeamCheckChunk; i < i1; ++i) {
                    const int e0 = row_ptr[2 * i], lb = row_ptr[2 * i + 1];
                    const int deg = row_ptr[2 * i + 2] - e0;
                    const double sigma = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0;
                    double *const M = lb >= 0 ? Lr + (size_t)lb * kTile : Mt + (size_t)e0 * kTile;   // (generic: LDS or the slot)
                    if (first) check_update<DC, true, true>(M, deg, sigma, r);     // (checks of up to 2 DC edges in two halves)
                    else check_update<DC, false, true>(M, deg, sigma, r);
                }
                return;
            }
            if (kTeamCheckChunk == 2 && !first && tp.pairs && i1 == c * 2 + 2) {
                // the usual case, two checks of the full degree: all 2 DC rows in flight at once
                const int i = c * 2;
                const int e0 = row_ptr[i], e1 = row_ptr[i + 1], e2 = row_ptr[i + 2];
                if (e1 - e0 == DC && e2 - e1 == DC) {
                    const double sg0 = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0, sg1 = ((syn[i + 1] >> lane) & 1ull) ? -1.0 : 1.0;
                    check_update_pair<DC>(Mt + (size_t)e0 * kTile, Mt + (size_t)e1 * kTile, sg0, sg1);
                    return;
                }
            }
            for (int i = c * kTeamCheckChunk; i < i1; ++i) {
                const int e0 = row_ptr[i];
                const int deg = row_ptr[i + 1] - e0;
                const double sigma = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0;
                if (first) check_update<DC, true>(Mt + (size_t)e0 * kTile, deg, sigma, r);
                else check_update<DC, false>(Mt + (size_t)e0 * kTile, deg, sigma, r);
            }
        };
        // this member's share is chunks rank, rank + G, ...: the first `stat` of them belong to its waves by right (chunk
        // l to wave l % W -- at least each wave's first), the rest they take from the counter in LDS
        const int mine = (nch - rank + G - 1) / G;
        const int stat = LROWS ? tp.rows.static_c : W;
        const int mirror = (LROWS && (tp.rows.flip & 1) && w >= W / 2) ? 2 * w + stat - W : -1;
        for (int l = w; l < mine;) {
            const int lc = (mirror >= 0 && l < stat) ? mirror - l : l;
            if (!(LROWS && lc < skip * W)) chunk(lc * G + rank);
            l = !tp.dynamic ? l + W : (l + W < stat ? l + W : stat + team_deal(&sh_deal[0], lane));
        }
    };
    // ... the first `pre` chunks of this wave's share of the NEXT check sweep, all of whose rows are on chip (TeamParams::pre)
    auto check_pre = [&](const int pre) {
        if constexpr (LROWS) {
            typedef int v4i __attribute__((ext_vector_type(4)));
            constexpr int FULL = (1 << DC) - 1;
            for (int k = 0; k < pre; ++k) {
                const int c = (w + k * W) * G + rank;
                for (int i = c * kTeamCheckChunk; i < (c + 1) * kTeamCheckChunk; ++i) {
                    const v4i ct = *(const v4i *)(col_ptr + 4 * i);
                    const double sg = ((syn[i] >> lane) & 1ull) ? -1.0 : 1.0;
                    if (RR > 0 && ct.z == FULL) check_update_regs<DC, false, TF>(ct.w, sg, r, rget, rput);
                    else check_update_exact<DC, false, TF>(Lr + (size_t)ct.y * kTile, sg, r);   // (ct.x == FULL: the host's promise, rows.first_c)
                }
            }
        }
    };
    // ---- variable-node sweep  (:152-178).  Across XCDs a chunk is 16 consecutive bits, so that a 128-byte line of
    //      decision words has one writer; inside one XCD (one L2) 4 bits, for an even finish
    auto var_sweep = [&](const bool first = false) {   // first: iteration 1 of a fresh tile without its check sweep (kFuseFirst)
        const int vb = (one_xcd || LROWS || IRR) ? 4 : 16;     // (LROWS / IRR: the host dealt the bits in chunks of 4)
        const int nch = (n + vb - 1) / vb;
        // LLR capture (tp.llr_raw 4 / 5, TeamParams): a store instruction costs the address path the same whatever its
        // width -- a 4-byte store per bit made the variable sweep 12 % longer, one 16-byte store per chunk of four
        // positions 3 %.  Position = place in the dealt bit order (LROWS), else the bit.
        const bool cap_on = WANT_LLR && (tp.llr_raw == 4 || tp.llr_raw == 5) && ((active >> lane) & 1ull);
        const size_t cap_rows = ((size_t)n + 3) & ~(size_t)3;
        unsigned int *const cap32 = (unsigned int *)p.llr + (size_t)tile * cap_rows * kTile;
        double *const cap64 = p.llr + (size_t)tile * cap_rows * kTile;
        auto cap_at = [&](int pos) { return ((size_t)(pos >> 2) * kTile + lane) * 4 + (size_t)(pos & 3); };
        auto capture1 = [&](int pos, double T) {
            if (!cap_on) return;
            if (tp.llr_raw == 4) cap32[cap_at(pos)] = llr_hi32(T);
            else cap64[cap_at(pos)] = T;
        };
        auto capture2 = [&](int pos, double T0, double T1) {      // pos even
            if (!cap_on) return;
            typedef unsigned int v2u __attribute__((ext_vector_type(2)));
            typedef double v2d __attribute__((ext_vector_type(2)));
            if (tp.llr_raw == 4) *(v2u *)(cap32 + cap_at(pos)) = v2u{llr_hi32(T0), llr_hi32(T1)};
            else *(v2d *)(cap64 + cap_at(pos)) = v2d{T0, T1};
        };
        auto capture4 = [&](int pos, const double (&T)[4]) {      // pos a multiple of 4
            if (!cap_on) return;
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            typedef double v2d __attribute__((ext_vector_type(2)));
            if (tp.llr_raw == 4) *(v4u *)(cap32 + cap_at(pos)) = v4u{llr_hi32(T[0]), llr_hi32(T[1]), llr_hi32(T[2]), llr_hi32(T[3])};
            else { *(v2d *)(cap64 + cap_at(pos)) = v2d{T[0], T[1]}; *(v2d *)(cap64 + cap_at(pos) + 2) = v2d{T[2], T[3]}; }
        };
        auto decide = [&](int pos, int j, double T, bool cap = true) {
            const u64 dec = __ballot(T >= 1.0);                                // :164-168
            if (WANT_LLR) {
                if (tp.llr_raw >= 4) { if (cap) capture1(pos, T); }
                else if (tp.llr_raw == 0 && ((active >> lane) & 1ull)) p.llr[((size_t)tile * n + j) * kTile + lane] = llr_of(T, p.llr_exact);  // :163
            }
            if (lane == 0) em[j] = dec;   // every lane; the stopped ones' decisions were captured when they stopped
        };
        auto chunk = [&](int c) {
            const int j1 = min(n, (c + 1) * vb);
            int j = c * vb;
            if constexpr (LROWS) {
                // positions of the dealt bit order; every bit has the full degree; csc2csr is vtab.  Everything the
                // table says about the four positions of a chunk is asked for at once (one scalar round trip per chunk)
                constexpr int VT = team_vtab_words(DV);
                if constexpr (kFuseFirst) {
                    if (first) {   // no rows to load: the messages of the first check sweep come from the table (bit_update_first)
                        if (j + 4 == j1) {
                            const int *const vt = csc2csr + (size_t)j * VT;
                            const TeamVRec<DV> A = team_vrec_load<DV>(vt), B = team_vrec_load<DV>(vt + VT), C = team_vrec_load<DV>(vt + 2 * VT),
                                               D = team_vrec_load<DV>(vt + 3 * VT);
                            double T[4];
                            T[0] = bit_update_first<DV, DC, TF>(Mt, Lr, sh_first, syn, lane, A.pos, A.lrow, r, rput);
                            T[1] = bit_update_first<DV, DC, TF>(Mt, Lr, sh_first, syn, lane, B.pos, B.lrow, r, rput);
                            T[2] = bit_update_first<DV, DC, TF>(Mt, Lr, sh_first, syn, lane, C.pos, C.lrow, r, rput);
                            T[3] = bit_update_first<DV, DC, TF>(Mt, Lr, sh_first, syn, lane, D.pos, D.lrow, r, rput);
                            decide(j, A.bit & 0x7fffffff, T[0], false);
                            decide(j + 1, B.bit & 0x7fffffff, T[1], false);
                            decide(j + 2, C.bit & 0x7fffffff, T[2], false);
                            decide(j + 3, D.bit & 0x7fffffff, T[3], false);
                            capture4(j, T);
                            return;
                        }
                        for (; j < j1; ++j) {
                            const TeamVRec<DV> A = team_vrec_load<DV>(csc2csr + (size_t)j * VT);
                            decide(j, A.bit & 0x7fffffff, bit_update_first<DV, DC, TF>(Mt, Lr, sh_first, syn, lane, A.pos, A.lrow, r, rput));
                        }
                        return;
                    }
                }
                auto single = [&](const TeamVRec<DV> &a, const int pa) {
                    if (a.bit >= 0) { decide(pa, a.bit, bit_update_exact_v<DV, TF>(Mt, a.pos, r)); return; }
                    if constexpr (RR > 0) {
                        int lo = a.lrow[0];
#pragma unroll
                        for (int k = 1; k < DV; ++k) lo = min(lo, a.lrow[k]);
                        if (lo <= -2) { decide(pa, a.bit & 0x7fffffff, bit_update_onchip<DV, TF>(Mt, Lr, Ldummy, a.pos, a.lrow, r, rget, rput)); return; }
                    }
                    decide(pa, a.bit & 0x7fffffff, bit_update_mixed<DV, TF>(Mt, Lr, a.pos, a.lrow, r));
                };
                auto two = [&](const TeamVRec<DV> &a, const TeamVRec<DV> &b, const int pa) {   // positions pa (even), pa + 1
                    int rest = -1;                            // stays -1: the edges 1 ... DV-1 of both are rows of the slot
#pragma unroll
                    for (int k = 1; k < DV; ++k) rest &= a.lrow[k] & b.lrow[k];
                    if ((a.bit | b.bit) >= 0 && tp.pairs) {   // neither has a row on chip: both loaded together
                        double T0, T1;
                        bit_update_pair_v<DV, TF>(Mt, a.pos, b.pos, r, T0, T1);
                        decide(pa, a.bit, T0, false);
                        decide(pa + 1, b.bit, T1, false);
                        capture2(pa, T0, T1);
                    } else if (rest == -1 && tp.pairs) {      // on chip at most the first edge of either (whole checks of the first block)
                        double T0, T1;
                        bit_update_pair_first<DV, TF>(Mt, Lr, a.pos, a.lrow[0], b.pos, b.lrow[0], r, rget, rput, T0, T1);
                        decide(pa, a.bit & 0x7fffffff, T0, false);
                        decide(pa + 1, b.bit & 0x7fffffff, T1, false);
                        capture2(pa, T0, T1);
                    } else {
                        single(a, pa);
                        single(b, pa + 1);
                    }
                };
                int q = j;
                if (LDPC_HOT(q + 4 == j1)) {                // the usual chunk: the table rows of all four positions at once
                    const int *const vt = csc2csr + (size_t)q * VT;
                    const TeamVRec<DV> A = team_vrec_load<DV>(vt), B = team_vrec_load<DV>(vt + VT), C = team_vrec_load<DV>(vt + 2 * VT),
                                       D = team_vrec_load<DV>(vt + 3 * VT);
                    if (LDPC_HOT(tp.pairs & 2)) {
                        int rest = -1;                        // stays -1: the edges 1 ... DV-1 of all four are rows of the slot
#pragma unroll
                        for (int k = 1; k < DV; ++k) rest &= A.lrow[k] & B.lrow[k] & C.lrow[k] & D.lrow[k];
                        if (LDPC_HOT(rest == -1)) {
                            int ps[4][DV];
#pragma unroll
                            for (int k = 0; k < DV; ++k) { ps[0][k] = A.pos[k]; ps[1][k] = B.pos[k]; ps[2][k] = C.pos[k]; ps[3][k] = D.pos[k]; }
                            const int lc[4] = {A.lrow[0], B.lrow[0], C.lrow[0], D.lrow[0]};
                            double T[4];
                            constexpr bool KINDS = LDPC_TEAM_CHUNK_KINDS < 0 ? DV == 3 : LDPC_TEAM_CHUNK_KINDS != 0;
                            const int lmin = min(min(lc[0], lc[1]), min(lc[2], lc[3])), lmax = max(max(lc[0], lc[1]), max(lc[2], lc[3]));
                            if (KINDS && RR > 0 && lmax <= -2) bit_update_multi_first<DV, 4, TF, 2>(Mt, Lr, ps, lc, r, rget, rput, T);
                            else if (KINDS && lmin >= 0) bit_update_multi_first<DV, 4, TF, 1>(Mt, Lr, ps, lc, r, rget, rput, T);
                            else bit_update_multi_first<DV, 4, TF, 0>(Mt, Lr, ps, lc, r, rget, rput, T);
                            decide(q, A.bit & 0x7fffffff, T[0], false);
                            decide(q + 1, B.bit & 0x7fffffff, T[1], false);
                            decide(q + 2, C.bit & 0x7fffffff, T[2], false);
                            decide(q + 3, D.bit & 0x7fffffff, T[3], false);
                            capture4(q, T);
                            return;
                        }
                    }
                    two(A, B, q);
                    two(C, D, q + 2);
                    return;
                }
                for (; q + 1 < j1; q += 2) {
                    const int *const vt = csc2csr + (size_t)q * VT;
                    const TeamVRec<DV> A = team_vrec_load<DV>(vt), B = team_vrec_load<DV>(vt + VT);
                    two(A, B, q);
                }
                for (; q < j1; ++q) single(team_vrec_load<DV>(csc2csr + (size_t)q * VT), q);
                return;
            }
            if constexpr (IRR) {
                for (; j < j1; ++j) {                          // j = position in the dealt order
                    const int c0 = col_ptr[2 * j], bw = col_ptr[2 * j + 1];
                    const int deg = col_ptr[2 * j + 2] - c0;
                    const double T = bw >= 0 ? bit_update<DV>(Mt, csc2csr + c0, deg, r) : bit_update_flat<DV>(Mt, Lr, csc2csr + c0, deg, r);
                    decide(j, bw & 0x7fffffff, T);
                }
                return;
            }
            for (; tp.pairs && j + 1 < j1; j += 2) {           // two bits of the full degree: all 2 DV rows in flight at once
                const int c0 = col_ptr[j], c1 = col_ptr[j + 1], c2 = col_ptr[j + 2];
                if (c1 - c0 != DV || c2 - c1 != DV) break;
                double T0, T1;
                bit_update_pair<DV>(Mt, csc2csr + c0, csc2csr + c1, r, T0, T1);
                decide(j, j, T0, false);
                decide(j + 1, j + 1, T1, false);
                capture2(j, T0, T1);
            }
            for (; j < j1; ++j) {
                const int c0 = col_ptr[j];
                const int deg = col_ptr[j + 1] - c0;
                decide(j, j, bit_update<DV>(Mt, csc2csr + c0, deg, r));
            }
        };
        const int mine = (nch - rank + G - 1) / G;
        const int stat = LROWS ? tp.rows.static_v : W;
        const int mirror = (LROWS && (tp.rows.flip & 2) && w >= W / 2) ? 2 * w + stat - W : -1;
        for (int l = w; l < mine;) {
            chunk(((mirror >= 0 && l < stat) ? mirror - l : l) * G + rank);
            l = !tp.dynamic ? l + W : (l + W < stat ? l + W : stat + team_deal(&sh_deal[1], lane));
        }
    };
    // ---- convergence test (:180-184) of iteration it_t on the decision words em_t: lane = check, words = 64 syndromes;
    //      the team ORs into mw[it_t - 1]
    auto run_test = [&](const int it_t, const u64 *em_t) {
        u64 mism = 0;
        // (a lane = a check, so s / 64 waves have work here -- at the C3 size half of a team's.  Counted member-minor: the
        //  first waves of EVERY member rather than all the waves of the first members, so that no member comes late to
        //  the sweep that follows as a whole -- its other waves take more of its dealt chunks meanwhile)
#if LDPC_TEAM_TEST_SPREAD
        const int gt = w * G + rank;
#else
        const int gt = gw;
#endif
        for (int i = gt * 64 + lane; i < s; i += GW * 64) {
            u64 par = 0;
            // (regular graphs of the rows-on-chip instantiations: check i's edges are rows DC i ... DC i + DC - 1 -- one
            //  dependent round trip less in front of the decision words)
            const int e1 = LROWS ? (i + 1) * DC : row_ptr[RS * (i + 1)];
            for (int e = LROWS ? i * DC : row_ptr[RS * i]; e < e1; e += 8) {
                int jb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) jb[q] = (e + q < e1) ? edge_bit[e + q] : -1;
                u64 wv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wv[q] = (jb[q] >= 0) ? em_t[jb[q]] : 0ull;
#pragma unroll
                for (int q = 0; q < 8; ++q) par ^= wv[q];
            }
            mism |= par ^ syn[i];
        }
        mism = wave_or(mism);
#if LDPC_TEAM_TEST_DIRECT
        // every wave tells the team by itself (an atomic nobody waits for; the next team barrier makes it visible):
        // gathering the member's words in LDS first cost a workgroup barrier in the middle of every iteration
        if (lane == 0 && mism) __hip_atomic_fetch_or(&mw[it_t - 1], mism, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (lane == 0) sh_mism[w] = mism;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 part = 0;
#pragma unroll
            for (int q = 0; q < W; ++q) part |= sh_mism[q];
            if (part) __hip_atomic_fetch_or(&mw[it_t - 1], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
    };
    // ---- the verdict of that test (after a team barrier): who stopped, and with which decisions
    auto verdict = [&](const int it_t, const u64 *em_t) {
        const u64 U = uniform64(never | __hip_atomic_load(&mw[it_t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const int total = it0 + it_t;                           // iterations of this lane's syndrome so far
        const u64 newly = active & ~U;
        if ((newly >> lane) & 1ull) { my_iters = total; my_conv = 1; }
        active &= U;
        const u64 spent = __ballot(total >= p.max_iters) & active;   // out of iterations: retires unconverged
        if ((spent >> lane) & 1ull) { my_iters = total; my_conv = 0; }
        active &= ~spent;
        // capture the stopping lanes' decisions of that iteration (every member its share of the bits; the decision
        // words were made visible by the barrier before the test, and nobody rewrites them before a team barrier
        // that this member only joins when it is through here: the next variable sweep writes the OTHER set of
        // words when the team runs ahead, and comes after the barrier of the next check sweep when it does not)
        const u64 stopped = newly | spent;
        if (stopped != 0) {
            for (int j = gw * 64 + lane; j < n; j += GW * 64) fin[j] = (fin[j] & ~stopped) | (em_t[j] & stopped);
        }
        quiet = stopped == 0;
        return total;
    };

    while (active != 0) {   // (every lane retires at the latest when its total reaches max_iters)
        ++it;
        em = (it & 1) ? em_odd : em_even;
        const u64 t0 = wall_clock64();
        // (kFuseFirst: a fresh tile's first iteration has no check sweep -- and, once the team knows where it runs, no barrier in
        //  front of its variable sweep: nothing that sweep reads was written by another member)
        const bool fused = kFuseFirst && it == 1 && !resumed;
        if (!have_check && !fused) check_sweep((it == 1) && !resumed);
        const u64 t1 = wall_clock64();
        if (!(fused && placed) && !team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;
        if (!placed) {
            if (threadIdx.x == 0)
                sh_one_xcd = __popc(__hip_atomic_load(xccs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1;
            __syncthreads();
            one_xcd = sh_one_xcd != 0 && !tp.always_release;
            placed = true;
        }
        if (have_check) {                                      // the test of iteration it - 1 rode with that sweep: its verdict
            have_check = false;
            verdict(it - 1, ((it - 1) & 1) ? em_odd : em_even);
            if (active == 0) break;                            // (every lane stopped: the sweep ahead was for nothing)
        }
        const u64 t2 = wall_clock64();
        var_sweep(fused);
        const u64 t3 = wall_clock64();
        // (everything the decision to run ahead depends on is known before the barrier: `quiet` is the last verdict's)
        const bool ahead = can_run_ahead && quiet && it >= tp.ahead_from && it + 2 <= horizon && it < p.max_iters && (int)__popcll(active) >= tp.ahead_min;
        const int pre = (LROWS && ahead) ? tp.pre : 0;
        if (pre > 0) {
            team_arrive(ctr, G, ++epoch, one_xcd);
            check_pre(pre);                                    // (on-chip checks of iteration it + 1: nobody else's rows)
            if (!team_wait(ctr, rank, epoch, tp.fault, tp.ticket, &sh_ok, sh_deal)) return;
        } else if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;
        run_test(it, em);
        if (ahead) {
            const u64 t3b = wall_clock64();
            check_sweep(false, pre);                           // iteration it + 1; the barrier at the top of the loop closes both
            have_check = true;
            const u64 t4 = wall_clock64();
            tk_check += (t1 - t0) + (t4 - t3b); tk_var += t3 - t2; tk_rest += (t2 - t1) + (t3b - t3);
            continue;
        }
        if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;
        const int total = verdict(it, em);
        const u64 t4 = wall_clock64();
        tk_check += t1 - t0; tk_var += t3 - t2; tk_rest += (t2 - t1) + (t4 - t3);   // rest = barriers + test
        // few stragglers left: hand them, with their messages, to the next level (decided alike by every member;
        // rank 0 reserves the room and tells the others through the team's control block, then all copy)
        if (p.defer_thresh != 0 && active != 0 && it >= p.defer_min_iter && (int)__popcll(active) <= p.defer_thresh) {
            if (LROWS || IRR) {   // the rows this member keeps in LDS go back to their places in the slot: the copy below reads the slot
                const int *const le = tp.rows.lds_edge + (size_t)rank * tp.rows.R;
                for (int q = w; q < tp.rows.R; q += W) {
                    const int e = le[q];
                    if (e >= 0) Mt[(size_t)e * kTile] = Lr[(size_t)q * kTile];
                }
                if constexpr (RR > 0) {   // ... and so do the rows this wave keeps in registers
                    const int *const re = tp.rows.reg_edge + (size_t)(rank * W + w) * tp.rows.regs;
                    for (int q = 0; q < tp.rows.regs; ++q) {
                        const int e = __builtin_amdgcn_readfirstlane(re[q]);
                        if (e >= 0) Mt[(size_t)e * kTile] = rget(q);
                    }
                }
            }
            if (rank == 0 && threadIdx.x == 0)
                __hip_atomic_store(ctr + 33, defer_reserve(cd->defer_count, (unsigned)__popcll(active), cd->next_cap) + 1u,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // base + 1; a full level (~0u) is told as 0
            if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;
            const unsigned told = __hip_atomic_load(ctr + 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;   // everyone has read it: the word may be rewritten
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
    horizon = it;
    if (rank == 0 && w == 0) {
        if (((valid & ~deferred) >> lane) & 1ull) {
            const long long ob = resumed ? (long long)cd->index[b0 + lane] : b0 + lane;
            cd->conv[ob] = (unsigned char)my_conv;
            if (cd->iters) cd->iters[ob] = my_iters;
            tot_iters += my_iters;
        }
    }
    // ---- the team's next tile.  The two barriers also keep a member from starting on the next tile (whose first
    //      check sweep overwrites the slot) while another still copies stragglers' rows out of it.
    if (ntiles <= nteams) break;                               // one tile per team: nothing to ask for
    if (rank == 0 && threadIdx.x == 0)
        __hip_atomic_store(ctr + 34, (unsigned)nteams + __hip_atomic_fetch_add(tile_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;
    tile = (int)min(__hip_atomic_load(ctr + 34, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (unsigned)ntiles);
    tile = __builtin_amdgcn_readfirstlane(tile);
    if (!team_barrier(ctr, G, rank, ++epoch, tp.fault, tp.ticket, &sh_ok, one_xcd, sh_deal)) return;   // everyone has read it
    }

    if (w == 0 && lane == 0) {   // diagnostics (LDPC_TEAM_DEBUG on the host): this member's own sweep times, where it ran
        unsigned int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned int *mine = ctr + kTeamCtlMember + 32 * rank;
        mine[1] = (unsigned int)tk_check; mine[2] = (unsigned int)tk_var; mine[3] = hw;
    }
    if (rank == 0 && w == 0) {
        int lo = (int)(tot_iters & 0xffffffffll), hi = (int)(tot_iters >> 32);   // (wave sum of a 64-bit count)
        u64 tot = 0;
        for (int l = 0; l < 64; ++l) tot += ((u64)(unsigned)__shfl(hi, l, 64) << 32) + (u64)(unsigned)__shfl(lo, l, 64);
        if (lane == 0) {
            atomicAdd(cd->sum_iters, tot);
            atomicAdd(&cd->phase_ticks[0], tk_check);
            atomicAdd(&cd->phase_ticks[1], tk_var);
            atomicAdd(&cd->phase_ticks[2], tk_rest);
        }
    }
}

}  // namespace ldpc
