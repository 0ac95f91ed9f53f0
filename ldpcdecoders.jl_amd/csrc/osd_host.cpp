// osd_host.cpp -- ordered-statistics post-processing of BP outputs, host side of
// libldpc_mi355x.so (BASELINE config 5: "BP+OSD post-processing on host").
//
// Replaces (QuantumSavory/LDPCDecoders.jl):
//   decode!(::BeliefPropagationOSDDecoder, syndrome) after its BP call
//                                     src/decoders/belief_propagation_osd.jl:52-60
//   osd(H, syndrome, bp_err, Val{0})  :63-125      osd(..., Val{O})  :127-209
//
// Same decisions as the reference, different mechanics: rows of H are 64-bit packed
// bitsets kept in the ORIGINAL column order; the reliability order is applied by
// visiting columns through the permutation instead of materialising H[:, perm]
// (row operations do not care about column order), the residual syndrome and the
// candidate weights are popcounts, and syndromes of a batch are spread over host threads.
#include "../../include/ldpc_mi355x.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace ldpc_detail {
ldpc_status set_error(ldpc_status st, const std::string &msg);  // defined in ldpc_mi355x.hip
}
using ldpc_detail::set_error;

typedef uint64_t u64;

struct ldpc_osd {
    int64_t m = 0, n = 0;
    int64_t order = 0;
    int64_t nw = 0;            // words per row
    std::vector<u64> rows;     // m * nw, bit c of row i = H[i, c]
};

namespace {

inline int parity_and(const u64 *a, const u64 *b, int64_t nw)
{
    u64 acc = 0;
    for (int64_t w = 0; w < nw; ++w) acc ^= a[w] & b[w];
    return __builtin_parityll(acc);
}
inline bool bit(const u64 *row, int64_t c) { return (row[c >> 6] >> (c & 63)) & 1u; }
inline void setbit(u64 *row, int64_t c, bool v)
{
    const u64 mask = (u64)1 << (c & 63);
    if (v) row[c >> 6] |= mask; else row[c >> 6] &= ~mask;
}

struct Work {
    std::vector<u64> W;        // working copy of the rows
    std::vector<u64> e, err, best, mrcmask, tmp;
    std::vector<uint8_t> s;    // transformed syndrome, one byte per row
    std::vector<double> key;
    std::vector<int64_t> perm, prow, pcol, mrc;
};

// one syndrome.  syn: m bytes (0/1), bp_err: n bytes, llr: n doubles, out: n bytes
void osd_one(const ldpc_osd &d, Work &k, const uint8_t *syn, const uint8_t *bp_err, const double *llr,
             uint8_t *out)
{
    const int64_t m = d.m, n = d.n, nw = d.nw;
    // reliability order (:53-55): key = max(exp(L), 1-exp(L)), stable, descending
    k.key.resize((size_t)n);
    k.perm.resize((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        const double p = std::exp(llr[j]);
        const double q = 1 - p;
        k.key[(size_t)j] = p > q ? p : q;
    }
    std::iota(k.perm.begin(), k.perm.end(), (int64_t)0);
    std::stable_sort(k.perm.begin(), k.perm.end(),
                     [&](int64_t a, int64_t b) { return k.key[(size_t)a] > k.key[(size_t)b]; });
    // BP hard decisions as a bitset (original column order)
    k.e.assign((size_t)nw, 0);
    for (int64_t j = 0; j < n; ++j)
        if (bp_err[j] == 1) k.e[(size_t)(j >> 6)] |= (u64)1 << (j & 63);
    k.s.resize((size_t)std::max<int64_t>(m, 1));

    if (d.order == 0) {
        // residual syndrome s_target = syndrome xor H*bp_err (:66-71)
        bool any = false;
        for (int64_t i = 0; i < m; ++i) {
            k.s[(size_t)i] = (uint8_t)((syn[i] != 0) ^ parity_and(&d.rows[(size_t)(i * nw)], k.e.data(), nw));
            any |= k.s[(size_t)i] != 0;
        }
        std::memcpy(out, bp_err, (size_t)n);
        if (!any) return;                                                    // :72-74
        k.W = d.rows;                                                        // H_work = copy(H)
        k.prow.clear(); k.pcol.clear();
        int64_t i = 0;
        for (int64_t jj = 0; jj < n; ++jj) {                                 // :81
            if (i >= m) break;
            bool rest = false;
            for (int64_t q = i; q < m && !rest; ++q) rest = k.s[(size_t)q] != 0;
            if (!rest) break;                                                // :82-84
            const int64_t c = k.perm[(size_t)jj];
            int64_t piv = -1;
            for (int64_t q = i; q < m; ++q)
                if (bit(&k.W[(size_t)(q * nw)], c)) { piv = q; break; }     // findfirst :86
            if (piv < 0) continue;
            if (bp_err[c] == 1)                                              // un-apply bp_err on the pivot column :88-90
                for (int64_t q = 0; q < m; ++q) k.s[(size_t)q] ^= (uint8_t)bit(&k.W[(size_t)(q * nw)], c);
            if (piv > i) {                                                   // :92-96
                std::swap_ranges(&k.W[(size_t)(i * nw)], &k.W[(size_t)(i * nw)] + nw, &k.W[(size_t)(piv * nw)]);
                std::swap(k.s[(size_t)i], k.s[(size_t)piv]);
            }
            const u64 *ri = &k.W[(size_t)(i * nw)];
            for (int64_t ii = i + 1; ii < m; ++ii) {                         // :98-103
                u64 *rr = &k.W[(size_t)(ii * nw)];
                if (bit(rr, c)) {
                    for (int64_t w = 0; w < nw; ++w) rr[w] ^= ri[w];
                    k.s[(size_t)ii] ^= k.s[(size_t)i];
                }
            }
            k.prow.push_back(i); k.pcol.push_back(c);
            ++i;
        }
        for (int64_t t = (int64_t)k.prow.size() - 1; t >= 0; --t) {          // back substitution :111-122
            const int64_t r = k.prow[(size_t)t], c = k.pcol[(size_t)t];
            out[c] = k.s[(size_t)r];
            if (out[c])
                for (int64_t ii = 0; ii < r; ++ii)
                    if (bit(&k.W[(size_t)(ii * nw)], c)) k.s[(size_t)ii] ^= 1;
        }
        return;
    }

    // ---- order > 0 (:127-209): full elimination, no shortcut
    k.W = d.rows;
    for (int64_t i = 0; i < m; ++i) k.s[(size_t)i] = syn[i];
    k.prow.clear(); k.pcol.clear();
    int64_t i = 0, jj = 0;
    while (i < m && jj < n) {                                                // :140-160
        const int64_t c = k.perm[(size_t)jj];
        int64_t piv = -1;
        for (int64_t q = i; q < m; ++q)
            if (bit(&k.W[(size_t)(q * nw)], c)) { piv = q; break; }
        if (piv < 0) { ++jj; continue; }
        if (piv > i) {
            std::swap_ranges(&k.W[(size_t)(i * nw)], &k.W[(size_t)(i * nw)] + nw, &k.W[(size_t)(piv * nw)]);
            std::swap(k.s[(size_t)i], k.s[(size_t)piv]);
        }
        const u64 *ri = &k.W[(size_t)(i * nw)];
        for (int64_t ii = i + 1; ii < m; ++ii) {
            u64 *rr = &k.W[(size_t)(ii * nw)];
            if (bit(rr, c)) {
                for (int64_t w = 0; w < nw; ++w) rr[w] ^= ri[w];
                k.s[(size_t)ii] ^= k.s[(size_t)i];
            }
        }
        k.prow.push_back(i); k.pcol.push_back(jj);   // pcol holds the SORTED position here
        ++i; ++jj;
    }
    const int64_t r = (int64_t)k.prow.size();
    for (int64_t t = r - 1; t >= 0; --t) {                                   // diagonalise :163-172
        const int64_t pi = k.prow[(size_t)t], c = k.perm[(size_t)k.pcol[(size_t)t]];
        const u64 *rp = &k.W[(size_t)(pi * nw)];
        for (int64_t ii = 0; ii < pi; ++ii) {
            u64 *rr = &k.W[(size_t)(ii * nw)];
            if (bit(rr, c)) {
                for (int64_t w = 0; w < nw; ++w) rr[w] ^= rp[w];
                k.s[(size_t)ii] ^= k.s[(size_t)pi];
            }
        }
    }
    int64_t order = d.order;
    if (order > n - r) order = n - r;                                        // :174-177 (the reference @warns)
    // most reliable (non-pivot) columns, ascending sorted position (:181)
    k.mrc.clear();
    k.mrcmask.assign((size_t)nw, 0);
    {
        size_t t = 0;
        for (int64_t pos = 0; pos < n; ++pos) {
            if (t < k.pcol.size() && k.pcol[t] == pos) { ++t; continue; }
            const int64_t c = k.perm[(size_t)pos];
            k.mrc.push_back(c);
            k.mrcmask[(size_t)(c >> 6)] |= (u64)1 << (c & 63);
        }
    }
    k.err = k.e;                                                             // :180
    k.best = k.e;                                                            // best_err = copy(bp_err) :179
    k.tmp.resize((size_t)nw);
    int64_t min_weight = n + 1;
    for (u64 x = 0; x < ((u64)1 << order); ++x) {                            // :184
        if (x != 0)
            for (int64_t q = 0; q < order; ++q) setbit(k.err.data(), k.mrc[(size_t)q], (x >> q) & 1u);  // :187-192
        for (int64_t w = 0; w < nw; ++w) k.tmp[(size_t)w] = k.err[(size_t)w] & k.mrcmask[(size_t)w];
        for (int64_t t = 0; t < r; ++t) {                                    // :194-199
            const int64_t pi = k.prow[(size_t)t], c = k.perm[(size_t)k.pcol[(size_t)t]];
            const bool v = (k.s[(size_t)pi] & 1u) ^ parity_and(&k.W[(size_t)(pi * nw)], k.tmp.data(), nw);
            setbit(k.err.data(), c, v);
        }
        int64_t weight = 0;
        for (int64_t w = 0; w < nw; ++w) weight += __builtin_popcountll(k.err[(size_t)w]);  // :200
        if (weight < min_weight) { min_weight = weight; k.best = k.err; }   // strict <: first minimum wins :202-205
    }
    for (int64_t j = 0; j < n; ++j) out[j] = (uint8_t)bit(k.best.data(), j);
}

}  // namespace

extern "C" {

ldpc_status ldpc_osd_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                            int64_t osd_order, ldpc_osd **out)
{
    if (!out) return set_error(LDPC_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    if (s < 0 || n < 0 || nnz < 0 || !colptr || (nnz > 0 && !rowval))
        return set_error(LDPC_ERR_INVALID_ARGUMENT, "bad dimensions or NULL pattern");
    if (osd_order < 0 || osd_order > 40) return set_error(LDPC_ERR_INVALID_ARGUMENT, "osd_order must be in [0, 40]");
    if (colptr[0] != 0 || colptr[n] != nnz) return set_error(LDPC_ERR_INVALID_ARGUMENT, "colptr is not a zero-based CSC pointer array");
    ldpc_osd *d = new (std::nothrow) ldpc_osd();
    if (!d) return set_error(LDPC_ERR_OUT_OF_MEMORY, "host allocation failed");
    d->m = s; d->n = n; d->order = osd_order; d->nw = (n + 63) / 64;
    d->rows.assign((size_t)(s * d->nw), 0);
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) { delete d; return set_error(LDPC_ERR_INVALID_ARGUMENT, "colptr is not non-decreasing"); }
        for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
            const int64_t i = rowval[q];
            if (i < 0 || i >= s) { delete d; return set_error(LDPC_ERR_INVALID_ARGUMENT, "rowval entry outside [0, s)"); }
            d->rows[(size_t)(i * d->nw + (j >> 6))] |= (u64)1 << (j & 63);
        }
    }
    *out = d;
    return LDPC_OK;
}

ldpc_status ldpc_osd_destroy(ldpc_osd *d)
{
    delete d;
    return LDPC_OK;
}

ldpc_status ldpc_osd_postprocess_batch(const ldpc_osd *d, int64_t batch, const uint8_t *syndromes,
                                       const uint8_t *bp_errors, const double *llr, uint8_t *errors,
                                       int32_t nthreads)
{
    if (!d) return set_error(LDPC_ERR_INVALID_ARGUMENT, "osd handle is NULL");
    if (batch < 0) return set_error(LDPC_ERR_INVALID_ARGUMENT, "negative batch");
    if (batch == 0) return LDPC_OK;
    if ((d->m > 0 && !syndromes) || (d->n > 0 && (!bp_errors || !llr || !errors)))
        return set_error(LDPC_ERR_INVALID_ARGUMENT, "NULL batch pointer");
    const size_t m = (size_t)d->m, n = (size_t)d->n;
    for (size_t q = 0; q < (size_t)batch * m; ++q)
        if (syndromes[q] > 1)  // Bool.(syndrome) is an InexactError in the reference (:66)
            return set_error(LDPC_ERR_INVALID_ARGUMENT, "OSD needs 0/1 syndrome entries");
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, batch));
    if (nthreads <= 0) {
        // auto: a thread is worth spawning for >= ~200 us of work (measured: 117 BB-72 syndromes on 117
        // threads of a 128-core host took 4.2 ms, on one thread 0.2 ms).  Word operations per syndrome:
        // elimination m*m*nw, reliability sort ~16 n, 2^w candidate patterns of m*nw each.
        const double w = (double)std::min<int64_t>(d->order, 20);
        const double per_syn = (double)d->m * (double)d->m * (double)d->nw + 16.0 * (double)d->n +
                               (d->order > 0 ? std::ldexp((double)d->m * (double)d->nw, (int)w) : 0.0);
        const double want = (double)batch * per_syn / 2.0e5;
        nt = (int)std::max(1.0, std::min((double)nt, want));
    }
    auto run = [&](int64_t lo, int64_t hi) {
        Work k;
        for (int64_t b = lo; b < hi; ++b)
            osd_one(*d, k, syndromes + (size_t)b * m, bp_errors + (size_t)b * n, llr + (size_t)b * n,
                    errors + (size_t)b * n);
    };
    if (nt == 1) {
        run(0, batch);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back(run, batch * t / nt, batch * (t + 1) / nt);
        for (auto &t : th) t.join();
    }
    return LDPC_OK;
}

}  // extern "C"
