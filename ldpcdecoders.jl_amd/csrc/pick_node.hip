// pick_node.hip -- instantiations of the node-parallel kernel (bp_node_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, int DV, bool LLR>
node_kernel_t node_pick_threads(int threads)
{
    return threads == 512 ? bp_node_kernel<DC, DV, LLR, 512> : bp_node_kernel<DC, DV, LLR, 1024>;
}
template <int DC, bool LLR>
node_kernel_t node_pick_dv(int dv, int threads)
{
    if (dv <= 4) return node_pick_threads<DC, 4, LLR>(threads);
    return node_pick_threads<DC, 16, LLR>(threads);
}
template <bool LLR>
node_kernel_t node_pick_dc(int dc, int dv, int threads)
{
    if (dc <= 8) return node_pick_dv<8, LLR>(dv, threads);
    if (dc <= 16) return node_pick_dv<16, LLR>(dv, threads);
    return node_pick_dv<32, LLR>(dv, threads);
}

}  // namespace

node_kernel_t pick_node_kernel(int dc, int dv, bool llr, int threads)
{
    return llr ? node_pick_dc<true>(dc, dv, threads) : node_pick_dc<false>(dc, dv, threads);
}

}  // namespace ldpc
