// pick_tile.hip -- instantiations of the HBM-streaming tile kernel (bp_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, int DV, bool LLR, bool SECOND>
bp_kernel_t pick_threads(int threads)
{
    switch (threads) {
    case 256: return bp_tile_kernel<DC, DV, LLR, 256, SECOND>;
    case 512: return bp_tile_kernel<DC, DV, LLR, 512, SECOND>;
    default: return bp_tile_kernel<DC, DV, LLR, 1024, SECOND>;
    }
}

template <int DC, bool LLR, bool SECOND>
bp_kernel_t pick_dv(int dv, int threads)
{
    if (dv <= 4) return pick_threads<DC, 4, LLR, SECOND>(threads);
    return pick_threads<DC, 16, LLR, SECOND>(threads);
}

template <bool LLR, bool SECOND>
bp_kernel_t pick_dc(int dc, int dv, int threads)
{
    if (dc <= 8) return pick_dv<8, LLR, SECOND>(dv, threads);
    if (dc <= 16) return pick_dv<16, LLR, SECOND>(dv, threads);
    return pick_dv<32, LLR, SECOND>(dv, threads);
}

}  // namespace

// This file is compiled twice (Makefile): LDPC_TILE_SECOND = 0 -> the first-pass instantiations and
// pick_kernel itself, 1 -> the instantiations of the straggler pass (<..., SECOND = true>).
#ifndef LDPC_TILE_SECOND
#define LDPC_TILE_SECOND 0
#endif
#if LDPC_TILE_SECOND
bp_kernel_t pick_kernel_second_pass(int dc, int dv, bool llr, int threads)
{
    return llr ? pick_dc<true, true>(dc, dv, threads) : pick_dc<false, true>(dc, dv, threads);
}
#else
bp_kernel_t pick_kernel_second_pass(int dc, int dv, bool llr, int threads);
bp_kernel_t pick_kernel(int dc, int dv, bool llr, int threads, bool second)
{
    if (second) return pick_kernel_second_pass(dc, dv, llr, threads);
    return llr ? pick_dc<true, false>(dc, dv, threads) : pick_dc<false, false>(dc, dv, threads);
}
#endif

}  // namespace ldpc
