/*
 * ldpc_mi355x_debug.h -- test hooks of libldpc_mi355x.so.  NOT part of the drop-in boundary (include/ldpc_mi355x.h):
 * nothing here replaces a reference interface; the CPU test-suite uses these entries to check, without a GPU, the
 * host-side planning the team kernel relies on (ldpcdecoders.jl_amd/csrc/ldpc_mi355x.hip: team_plan_pure(),
 * team_rows_tables()).  Pure host code; no device is needed or touched.
 */
#ifndef LDPC_MI355X_DEBUG_H
#define LDPC_MI355X_DEBUG_H

#include "ldpc_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* How a batch of `batch` syndromes of a graph with nnz edges would be dealt to teams of workgroups on an MI355X
   (256 CUs, one team workgroup per CU, members of >= 2048 message rows) under a budget of cache_mib MiB of message
   slots in flight (the library's default: 240).  rows_dv: the bit degree when the graph is regular and has a
   rows-on-chip instantiation, else 0 (with it the plan counts on 312 rows in LDS and 8 x 32 in registers per member).  out = { members per team (1 = no teams: node-parallel or tile kernel),
   teams = message slots in flight, workgroups launched, XCDs that host teams, 1 if the members of a team are dealt
   over all XCDs (<= 3 tiles), 1 if members keep rows in LDS }. */
ldpc_status ldpc_debug_team_plan(int64_t nnz, int64_t max_iters, int64_t batch, int32_t cache_mib, int32_t rows_dv,
                                 int32_t out[6]);

/* The tables with which the team kernel keeps message rows ON CHIP -- in the members' LDS and in the registers of
   their waves -- for a REGULAR graph (every check dc edges, every bit dv) whose degree pair has an instantiation --
   check degree 6 ... 10 x bit degree 3 ... 5; LDPC_ERR_UNSUPPORTED otherwise -- and teams of `members` workgroups of 8 waves.
   In: regs_per_wave (0 ... 32) rows a wave may keep in registers; static_quarters (0 ... 4): the share of a member's
   chunks per sweep that its waves own by right (only edges between such chunks of ONE wave can live in its registers).
   Out: degrees = {dc, dv}; shape = {words per position record (8 or 16), R = LDS rows per member (at most 312),
   static check chunks, static position chunks per member, register rows per wave in effect};
   vtab [n][shape[0]] = per position of the dealt bit order the CSR rows of its dv edges (row = dc * check + place among
   the check's bits), where each lives (>= 0: that LDS row of the member; -1: the team's slot; <= -2: register row
   -2 - x of the wave), the bit (| 1 << 31 when one of its edges is not in the slot), padding; ctab [s][4] = per check
   the mask of its edges in LDS, the LDS row of the first of them, the mask of its edges in registers, the register
   row of the first of them; lds_edge [members][R] and reg_edge [members][8][regs_per_wave] = the CSR rows held, -1
   beyond the count (pass room for members * 312 and members * 8 * 32).
   No reference counterpart: the reference keeps every message in one dense matrix (belief_propagation.jl:83-91). */
ldpc_status ldpc_debug_team_rows(int64_t s, int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t members,
                                 int32_t regs_per_wave, int32_t static_quarters, int32_t *degrees, int32_t *shape,
                                 int32_t *vtab, int32_t *ctab, int32_t *lds_edge, int32_t *reg_edge);

/* The tables with which the team kernel keeps WHOLE CHECKS of an IRREGULAR graph (any CSC pattern) in the LDS of their
   owners (csrc/ldpc_mi355x.hip team_irr_tables(); bp_team_kernels.hpp, IRR), for teams of `members` workgroups and nodes
   inside the register buckets dc_bucket / dv_bucket.  Out: shape = {R = LDS rows per member (at most 312), rows in LDS in
   all}; ctab2 [s + 1][2] = per check {its first CSR row, its first LDS row or -1}; ptab [n + 1][2] = per position of the
   dealt bit order {first entry of its edge list in ploc, the bit | 1 << 31 when one of its edges is in LDS}; ploc [nnz] =
   per edge of a position (checks ascending) its CSR row, or -1 - (LDS row); lds_edge [members][R] (pass room for
   members * 312) = the CSR rows held, -1 beyond a member's count; posmap [n] = the position of every bit.
   No reference counterpart. */
ldpc_status ldpc_debug_team_irr(int64_t s, int64_t n, const int64_t *colptr, const int64_t *rowval, int32_t members,
                                int32_t dc_bucket, int32_t dv_bucket, int32_t *shape, int32_t *ctab2, int32_t *ptab, int32_t *ploc,
                                int32_t *lds_edge, int32_t *posmap);

/* Two builds of the library in one process (the product and the -DLDPC_EXPERIMENTS build: the Python host of the tests)
   must not run team grids on one device at the same time -- every member of a team has to be resident.  Each build
   orders its own team launches through a per-device event table; the build loaded second adopts the table of the
   first: adopt(process_state of the other build).  Call before the adopting build has launched anything. */
void *ldpc_debug_process_state(void);
ldpc_status ldpc_debug_adopt_process_state(void *state);

/* The one entry here that needs the GPU: out_core[i] = div_core(num[i], den[i]) -- the nine instructions in the middle of
   hipcc's IEEE double division, which the check sweep runs alone where a wave's operands rule the end cases out
   (bp_kernels.hpp, LDPC_FAST_DIV) -- and out_ieee[i] = num[i] / den[i], computed on the device from HOST arrays of
   `count` doubles each.  A GPU test holds the two against each other over the ranges the kernels take the short form
   in, edges included.  No reference counterpart (Julia's `/` is the IEEE division). */
ldpc_status ldpc_debug_div_check(int64_t count, const double *num, const double *den, double *out_core, double *out_ieee);

/* ... and one more: out_fast[i] = the LLR every kernel of the library returns for posterior odds odds[i] (bp_kernels.hpp
   llr_of without llr_exact: the odds cut to their upper 32 bits, then llr_cut -- frexp, one division, seven terms of
   2 atanh), out_lib[i] = the library's log(1 / .) of the same cut odds; HOST arrays of `count` doubles.  A GPU test holds
   them against each other (<= 1e-12, end cases identical) and against log(1 / odds) (<= 5e-7).  Reference:
   belief_propagation.jl:163. */
ldpc_status ldpc_debug_llr_check(int64_t count, const double *odds, double *out_fast, double *out_lib);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_MI355X_DEBUG_H */
