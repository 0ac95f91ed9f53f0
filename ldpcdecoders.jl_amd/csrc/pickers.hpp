// pickers.hpp -- the kernel instantiations live in translation units of their own (pick_*.hip), so that
// they compile in parallel and an edit to one kernel family does not rebuild the others; the host code
// (ldpc_mi355x.hip) only sees these selectors.  Each returns the instantiation for the smallest register
// bucket that holds the graph's maximum check / bit degree.
#pragma once
#include "bp_kernels.hpp"
#include "bp_lds_kernels.hpp"
#include "bp_node_kernels.hpp"
#include "bp_team_kernels.hpp"

namespace ldpc {

typedef void (*bp_kernel_t)(BPParams, const int *, const int *, const int *, const int *, const u64 *, const u64 *);
typedef void (*lds_kernel_t)(LdsParams, const int *, const int *, const int *, const int *);
typedef void (*node_kernel_t)(NodeParams, const int *, const int *, const int *, const int *);
typedef void (*team_kernel_t)(BPParams, TeamParams, const int *, const int *, const int *, const int *, const u64 *,
                              const u64 *);

bp_kernel_t pick_kernel(int dc, int dv, bool llr, int threads, bool second = false);   // pick_tile.hip
lds_kernel_t pick_lds_kernel(int dc, int dv, bool llr, int threads);                   // pick_lds.hip
node_kernel_t pick_node_kernel(int dc, int dv, bool llr, int threads, int msg);   // pick_node.hip; msg: bp_node_kernels.hpp
team_kernel_t pick_team_kernel(int dc, int dv, bool llr, bool resumed = false);        // pick_team.hip
team_kernel_t pick_team_kernel_rows(int dc, int dv, bool llr, bool regs);   // pick_team.hip: regular graphs of these exact degrees, rows in LDS (nullptr: none)
team_kernel_t pick_team_kernel_irr(int dc, int dv, bool llr);              // pick_team.hip: irregular graphs, whole checks in LDS (by register bucket)
// the register buckets of the team kernels: nodes up to this degree are straight-line code (wider ones: the O(deg^2) path)
inline int team_bucket_dc(int dc) { return dc <= 8 ? 8 : dc <= 16 ? 16 : 32; }
inline int team_bucket_dv(int dv) { return dv <= 4 ? 4 : 16; }

}  // namespace ldpc
