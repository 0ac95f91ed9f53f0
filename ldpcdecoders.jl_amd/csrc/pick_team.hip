// pick_team.hip -- instantiations of the team kernel (bp_team_kernels.hpp) (see pickers.hpp).
// Message rows are stored WRITE-THROUGH and agent-coherent (sc1) in every team kernel: members of a team over several XCDs
// (wide teams, a single decode!) read each other's rows, and with write-back stores the release side of every team barrier is a
// write-back of a whole L2 (buffer_wbl2) -- n = 65536, 16,384 syndromes x 50 iterations: 885 -> 840 ms, n = 32768: 429 -> 415 ms;
// one-XCD teams measure the same either way (C3 full-50 701.5 against 701.2 ms).  profiles/r04_wide_teams.txt.
#ifndef LDPC_STM_SC1
#define LDPC_STM_SC1 1
#endif
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, bool LLR, bool RESUMED>
team_kernel_t team_pick_dv(int dv)
{
    if (dv <= 4) return bp_team_kernel<DC, 4, LLR, LDPC_TEAM_THREADS, RESUMED>;
    return bp_team_kernel<DC, 16, LLR, LDPC_TEAM_THREADS, RESUMED>;
}
template <bool LLR, bool RESUMED>
team_kernel_t team_pick_dc(int dc, int dv)
{
    if (dc <= 8) return team_pick_dv<8, LLR, RESUMED>(dv);
    if (dc <= 16) return team_pick_dv<16, LLR, RESUMED>(dv);
    return team_pick_dv<32, LLR, RESUMED>(dv);
}

}  // namespace

// This file is compiled twice (Makefile): LDPC_TEAM_RESUMED = 0 -> the instantiations for fresh tiles and
// pick_team_kernel itself, 1 -> the instantiations of the passes over the packed levels (<..., RESUMED = true>).
#ifndef LDPC_TEAM_RESUMED
#define LDPC_TEAM_RESUMED 0
#endif
#ifndef LDPC_TEAM_ROWS
#define LDPC_TEAM_ROWS 0
#endif
#ifndef LDPC_TEAM_IRR
#define LDPC_TEAM_IRR 0
#endif
#if LDPC_TEAM_IRR
// (one more compilation: -DLDPC_TEAM_IRR=1) irregular graphs with whole checks in LDS (bp_team_kernels.hpp, IRR), by register
// bucket like the plain team kernel
namespace {
template <int DC, bool LLR>
team_kernel_t irr_pick_dv(int dv)
{
    if (dv <= 4) return bp_team_kernel<DC, 4, LLR, LDPC_TEAM_THREADS, false, false, 0, true>;
    return bp_team_kernel<DC, 16, LLR, LDPC_TEAM_THREADS, false, false, 0, true>;
}
template <bool LLR>
team_kernel_t irr_pick_dc(int dc, int dv)
{
    if (dc <= 8) return irr_pick_dv<8, LLR>(dv);
    if (dc <= 16) return irr_pick_dv<16, LLR>(dv);
    return nullptr;   // (the 32-wide bucket on generic pointers spills: the host asks for 8 or 16 -- team_irr_dc_bucket(), ldpc_mi355x.hip)
}
}  // namespace
team_kernel_t pick_team_kernel_irr(int dc, int dv, bool llr) { return llr ? irr_pick_dc<true>(dc, dv) : irr_pick_dc<false>(dc, dv); }
#elif LDPC_TEAM_ROWS
// (further compilations: -DLDPC_TEAM_ROWS=1 -DLDPC_TEAM_ROWS_DC=6 ... 10, one object per check degree so that they build in
// parallel) rows on chip: regular graphs, EXACT degrees (team_rows_degrees_ok(): check degree 6 ... 10 x bit degree
// 3 ... 5, north_star's "row-weight ~6-10"), fresh tiles.  nullptr: no instantiation for this pair.
#ifndef LDPC_TEAM_ROWS_DC
#error "compile with -DLDPC_TEAM_ROWS_DC=<check degree>"
#endif
namespace {
template <int DC, int DV>
team_kernel_t rows_pick(bool llr, bool regs)
{
    static_assert(team_rows_degrees_ok(DC, DV), "keep team_rows_degrees_ok() and the instantiations in step");
    if (regs) {
        if (llr) return bp_team_kernel<DC, DV, true, LDPC_TEAM_THREADS, false, true, kTeamRegRows>;
        return bp_team_kernel<DC, DV, false, LDPC_TEAM_THREADS, false, true, kTeamRegRows>;
    }
    if (llr) return bp_team_kernel<DC, DV, true, LDPC_TEAM_THREADS, false, true>;
    return bp_team_kernel<DC, DV, false, LDPC_TEAM_THREADS, false, true>;
}
}  // namespace
// regs: the instantiation whose waves also keep rows in registers (TeamRows::regs > 0)
#define LDPC_ROWS_PICK_NAME2(dc) pick_team_kernel_rows_dc##dc
#define LDPC_ROWS_PICK_NAME(dc) LDPC_ROWS_PICK_NAME2(dc)
team_kernel_t LDPC_ROWS_PICK_NAME(LDPC_TEAM_ROWS_DC)(int dv, bool llr, bool regs)
{
    if (dv == 3) return rows_pick<LDPC_TEAM_ROWS_DC, 3>(llr, regs);
    if (dv == 4) return rows_pick<LDPC_TEAM_ROWS_DC, 4>(llr, regs);
    if (dv == 5) return rows_pick<LDPC_TEAM_ROWS_DC, 5>(llr, regs);
    return nullptr;
}
#elif LDPC_TEAM_RESUMED
team_kernel_t pick_team_kernel_resumed(int dc, int dv, bool llr)
{
    return llr ? team_pick_dc<true, true>(dc, dv) : team_pick_dc<false, true>(dc, dv);
}
#else
team_kernel_t pick_team_kernel_resumed(int dc, int dv, bool llr);
team_kernel_t pick_team_kernel_rows_dc6(int dv, bool llr, bool regs);
team_kernel_t pick_team_kernel_rows_dc7(int dv, bool llr, bool regs);
team_kernel_t pick_team_kernel_rows_dc8(int dv, bool llr, bool regs);
team_kernel_t pick_team_kernel_rows_dc9(int dv, bool llr, bool regs);
team_kernel_t pick_team_kernel_rows_dc10(int dv, bool llr, bool regs);
team_kernel_t pick_team_kernel_rows(int dc, int dv, bool llr, bool regs)
{
    if (!team_rows_degrees_ok(dc, dv)) return nullptr;
    switch (dc) {
    case 6: return pick_team_kernel_rows_dc6(dv, llr, regs);
    case 7: return pick_team_kernel_rows_dc7(dv, llr, regs);
    case 8: return pick_team_kernel_rows_dc8(dv, llr, regs);
    case 9: return pick_team_kernel_rows_dc9(dv, llr, regs);
    case 10: return pick_team_kernel_rows_dc10(dv, llr, regs);
    }
    return nullptr;
}
team_kernel_t pick_team_kernel(int dc, int dv, bool llr, bool resumed)
{
    if (resumed) return pick_team_kernel_resumed(dc, dv, llr);
    return llr ? team_pick_dc<true, false>(dc, dv) : team_pick_dc<false, false>(dc, dv);
}
#endif

}  // namespace ldpc
