// pick_team.hip -- instantiations of the team kernel (bp_team_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, bool LLR>
team_kernel_t team_pick_dv(int dv)
{
    if (dv <= 4) return bp_team_kernel<DC, 4, LLR, 512>;
    return bp_team_kernel<DC, 16, LLR, 512>;
}
template <bool LLR>
team_kernel_t team_pick_dc(int dc, int dv)
{
    if (dc <= 8) return team_pick_dv<8, LLR>(dv);
    if (dc <= 16) return team_pick_dv<16, LLR>(dv);
    return team_pick_dv<32, LLR>(dv);
}

}  // namespace

team_kernel_t pick_team_kernel(int dc, int dv, bool llr)
{
    return llr ? team_pick_dc<true>(dc, dv) : team_pick_dc<false>(dc, dv);
}

}  // namespace ldpc
