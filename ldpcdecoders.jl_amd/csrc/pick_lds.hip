// pick_lds.hip -- instantiations of the LDS-resident kernel (bp_lds_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, int DV, bool LLR>
lds_kernel_t lds_pick_threads(int threads)
{
    switch (threads) {
    case 256: return bp_lds_kernel<DC, DV, LLR, 256>;
    case 512: return bp_lds_kernel<DC, DV, LLR, 512>;
    default: return bp_lds_kernel<DC, DV, LLR, 1024>;
    }
}
template <int DC, bool LLR>
lds_kernel_t lds_pick_dv(int dv, int threads)
{
    if (dv <= 4) return lds_pick_threads<DC, 4, LLR>(threads);
    return lds_pick_threads<DC, 16, LLR>(threads);
}
template <bool LLR>
lds_kernel_t lds_pick_dc(int dc, int dv, int threads)
{
    if (dc <= 8) return lds_pick_dv<8, LLR>(dv, threads);
    if (dc <= 16) return lds_pick_dv<16, LLR>(dv, threads);
    return lds_pick_dv<32, LLR>(dv, threads);
}

}  // namespace

lds_kernel_t pick_lds_kernel(int dc, int dv, bool llr, int threads)
{
    return llr ? lds_pick_dc<true>(dc, dv, threads) : lds_pick_dc<false>(dc, dv, threads);
}

}  // namespace ldpc
