// pick_node.hip -- instantiations of the node-parallel kernel (bp_node_kernels.hpp) (see pickers.hpp).
#include "pickers.hpp"

namespace ldpc {

namespace {
template <int DC, int DV, bool LLR>
node_kernel_t node_pick_threads(int threads, bool msg_lds)
{
    if (msg_lds) return bp_node_kernel<DC, DV, LLR, 1024, true>;   // one 16-wave workgroup owns the CU's LDS
    return threads == 512 ? bp_node_kernel<DC, DV, LLR, 512, false> : bp_node_kernel<DC, DV, LLR, 1024, false>;
}
template <int DC, bool LLR>
node_kernel_t node_pick_dv(int dv, int threads, bool msg_lds)
{
    if (dv <= 4) return node_pick_threads<DC, 4, LLR>(threads, msg_lds);
    return node_pick_threads<DC, 16, LLR>(threads, msg_lds);
}
template <bool LLR>
node_kernel_t node_pick_dc(int dc, int dv, int threads, bool msg_lds)
{
    if (dc <= 8) return node_pick_dv<8, LLR>(dv, threads, msg_lds);
    if (dc <= 16) return node_pick_dv<16, LLR>(dv, threads, msg_lds);
    return node_pick_dv<32, LLR>(dv, threads, msg_lds);
}

}  // namespace

node_kernel_t pick_node_kernel(int dc, int dv, bool llr, int threads, bool msg_lds)
{
    return llr ? node_pick_dc<true>(dc, dv, threads, msg_lds) : node_pick_dc<false>(dc, dv, threads, msg_lds);
}

}  // namespace ldpc
