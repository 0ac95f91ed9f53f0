// pick_node.hip -- instantiations of the node-parallel kernel (bp_node_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, int DV, bool LLR>
node_kernel_t node_pick_threads(int threads, int msg)
{
    if (msg == 1) return bp_node_kernel<DC, DV, LLR, 1024, 1>;   // one 16-wave workgroup owns the CU's LDS
    if (msg == 2) return bp_node_kernel<DC, DV, LLR, 1024, 2>;
    return threads == 512 ? bp_node_kernel<DC, DV, LLR, 512, 0> : bp_node_kernel<DC, DV, LLR, 1024, 0>;
}
template <int DC, bool LLR>
node_kernel_t node_pick_dv(int dv, int threads, int msg)
{
    if (dv <= 4) return node_pick_threads<DC, 4, LLR>(threads, msg);
    return node_pick_threads<DC, 16, LLR>(threads, msg);
}
template <bool LLR>
node_kernel_t node_pick_dc(int dc, int dv, int threads, int msg)
{
    if (dc <= 8) return node_pick_dv<8, LLR>(dv, threads, msg);
    if (dc <= 16) return node_pick_dv<16, LLR>(dv, threads, msg);
    return node_pick_dv<32, LLR>(dv, threads, msg);
}

}  // namespace

node_kernel_t pick_node_kernel(int dc, int dv, bool llr, int threads, int msg)
{
    return llr ? node_pick_dc<true>(dc, dv, threads, msg) : node_pick_dc<false>(dc, dv, threads, msg);
}

}  // namespace ldpc
