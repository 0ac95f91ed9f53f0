/*
 * osd_oracle.c -- CPU ORACLE for the BP+OSD host step (test infrastructure, NOT product code).
 *
 * Literal, dense (one byte per matrix entry) restatement of
 *   decode!(::BeliefPropagationOSDDecoder, syndrome)   src/decoders/belief_propagation_osd.jl:49-61
 *   rowswap!                                            :31-36
 *   osd(H, syndrome, bp_err, Val{0})                    :63-125
 *   osd(H, syndrome, bp_err, Val{O})                    :127-209
 * given the BP outputs (hard decisions + log_probabs).  PARITY UNPINNED: no Julia runtime,
 * no golden vectors in the reference; pinned by the reference's own properties
 * (test/test_bposd_decoder.jl: exact recovery at per=0.01, syndrome consistency always).
 * `exp` is libm's (Julia's Base.exp is a different <=1-ulp implementation): the column
 * order can in principle differ from Julia's for keys within an ulp of each other.
 *
 * Deliberately shares no code with the bit-packed product implementation
 * (ldpcdecoders.jl_amd/csrc/osd_host.cpp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* stable merge sort of indices, DESCENDING by key, ties keep ascending index
 * (sortperm(..., rev=true), belief_propagation_osd.jl:55) */
static void merge_sort_desc(const double *key, int64_t *idx, int64_t *tmp, int64_t lo, int64_t hi)
{
    if (hi - lo < 2) return;
    int64_t mid = lo + (hi - lo) / 2;
    merge_sort_desc(key, idx, tmp, lo, mid);
    merge_sort_desc(key, idx, tmp, mid, hi);
    int64_t a = lo, b = mid, o = lo;
    while (a < mid && b < hi) {
        /* take from the right run only if strictly greater: equal keys keep left (lower index) first */
        if (key[idx[b]] > key[idx[a]]) tmp[o++] = idx[b++];
        else tmp[o++] = idx[a++];
    }
    while (a < mid) tmp[o++] = idx[a++];
    while (b < hi) tmp[o++] = idx[b++];
    memcpy(idx + lo, tmp + lo, sizeof(int64_t) * (size_t)(hi - lo));
}

static void rowswap(uint8_t *H, int64_t n, int64_t i, int64_t j)            /* :31-36 */
{
    if (i == j) return;
    for (int64_t c = 0; c < n; ++c) {
        uint8_t t = H[i * n + c]; H[i * n + c] = H[j * n + c]; H[j * n + c] = t;
    }
}

/* osd(H, syndrome, bp_err, Val{0})  :63-125.  H is m x n row-major (already column-sorted). */
static void osd0(const uint8_t *H, int64_t m, int64_t n, const uint8_t *syndrome, const uint8_t *bp_err,
                 uint8_t *out)
{
    uint8_t *s_target = (uint8_t *)malloc((size_t)(m ? m : 1));
    for (int64_t i = 0; i < m; ++i) s_target[i] = syndrome[i] != 0;        /* Bool.(syndrome) :66 */
    for (int64_t j = 0; j < n; ++j)                                         /* :67-71 */
        if (bp_err[j] == 1)
            for (int64_t i = 0; i < m; ++i) s_target[i] ^= H[i * n + j];
    int any = 0;
    for (int64_t i = 0; i < m; ++i) any |= s_target[i];
    if (!any) { memcpy(out, bp_err, (size_t)n); free(s_target); return; }   /* :72-74 */

    uint8_t *W = (uint8_t *)malloc((size_t)((m > 0 && n > 0) ? m * n : 1));            /* H_work = copy(H) :76 */
    memcpy(W, H, (size_t)(m * n));
    int64_t *prow = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    int64_t *pcol = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    int64_t np = 0, i = 0;
    for (int64_t j = 0; j < n; ++j) {                                       /* :81 */
        int rest = 0;
        for (int64_t q = i; q < m; ++q) rest |= s_target[q];
        if (i >= m || !rest) break;                                         /* :82-84 */
        int64_t k = -1;
        for (int64_t q = i; q < m; ++q) if (W[q * n + j]) { k = q; break; } /* findfirst :86 */
        if (k >= 0) {
            if (bp_err[j] == 1)                                             /* :88-90 */
                for (int64_t q = 0; q < m; ++q) s_target[q] ^= W[q * n + j];
            if (k > i) {                                                    /* :92-96 */
                rowswap(W, n, i, k);
                uint8_t t = s_target[i]; s_target[i] = s_target[k]; s_target[k] = t;
            }
            for (int64_t ii = i + 1; ii < m; ++ii)                          /* :98-103 */
                if (W[ii * n + j]) {
                    for (int64_t c = 0; c < n; ++c) W[ii * n + c] ^= W[i * n + c];
                    s_target[ii] ^= s_target[i];
                }
            prow[np] = i; pcol[np] = j; ++np;                               /* :105-106 */
            ++i;
        }
    }
    memcpy(out, bp_err, (size_t)n);                                         /* correction :110 */
    for (int64_t t = np - 1; t >= 0; --t) {                                 /* :111-122 */
        int64_t r = prow[t], c = pcol[t];
        out[c] = s_target[r];
        if (out[c])
            for (int64_t ii = 0; ii < r; ++ii)
                if (W[ii * n + c]) s_target[ii] ^= 1;
    }
    free(s_target); free(W); free(prow); free(pcol);
}

/* osd(H, syndrome, bp_err, Val{O})  :127-209.  H (m x n row-major) is modified in place. */
static void osdw(uint8_t *H, int64_t m, int64_t n, const uint8_t *syndrome, const uint8_t *bp_err,
                 int64_t osd_order, uint8_t *out)
{
    int64_t *prow = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    int64_t *pcol = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    uint8_t *s = (uint8_t *)malloc((size_t)(m ? m : 1));
    memcpy(s, syndrome, (size_t)m);                                         /* :138 */
    int64_t r = 0, i = 0, j = 0;
    while (i < m && j < n) {                                                /* :140-160 */
        int64_t k = -1;
        for (int64_t q = i; q < m; ++q) if (H[q * n + j]) { k = q; break; }
        if (k < 0) { ++j; continue; }
        if (k > i) {
            rowswap(H, n, i, k);
            uint8_t t = s[i]; s[i] = s[k]; s[k] = t;
        }
        for (int64_t ii = i + 1; ii < m; ++ii)
            if (H[ii * n + j]) {
                for (int64_t c = 0; c < n; ++c) H[ii * n + c] ^= H[i * n + c];
                s[ii] ^= s[i];
            }
        prow[r] = i; pcol[r] = j;
        ++i; ++j; ++r;
    }
    for (int64_t t = r - 1; t >= 0; --t) {                                  /* :163-172 */
        int64_t pi = prow[t], pj = pcol[t];
        for (int64_t ii = 0; ii < pi; ++ii)
            if (H[ii * n + pj]) {
                for (int64_t c = 0; c < n; ++c) H[ii * n + c] ^= H[pi * n + c];
                s[ii] ^= s[pi];
            }
    }
    if (osd_order > n - r) osd_order = n - r;                               /* :174-177 (@warn) */
    uint8_t *err = (uint8_t *)malloc((size_t)(n ? n : 1));
    memcpy(err, bp_err, (size_t)n);                                         /* :180 */
    memcpy(out, bp_err, (size_t)n);                                         /* best_err :179 */
    uint8_t *is_pivot = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    for (int64_t t = 0; t < r; ++t) is_pivot[pcol[t]] = 1;
    int64_t nm = 0, *mrc = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    for (int64_t c = 0; c < n; ++c) if (!is_pivot[c]) mrc[nm++] = c;        /* setdiff(1:n, cols) :181 */
    int64_t min_weight = n + 1;
    for (uint64_t x = 0; x < ((uint64_t)1 << osd_order); ++x) {             /* :184 */
        if (x != 0)                                                         /* :187-192 */
            for (int64_t q = 0; q < osd_order; ++q) err[mrc[q]] = (uint8_t)((x >> q) & 1u);
        for (int64_t t = 0; t < r; ++t) {                                   /* :194-199 */
            uint8_t v = s[prow[t]];
            for (int64_t q = 0; q < nm; ++q) v ^= (uint8_t)(H[prow[t] * n + mrc[q]] & err[mrc[q]]);
            err[pcol[t]] = v;
        }
        int64_t weight = 0;
        for (int64_t c = 0; c < n; ++c) weight += err[c];                   /* :200 */
        if (weight < min_weight) { min_weight = weight; memcpy(out, err, (size_t)n); }  /* :202-205 */
    }
    free(prow); free(pcol); free(s); free(err); free(is_pivot); free(mrc);
}

/* decode!(::BeliefPropagationOSDDecoder) after the BP call: :52-60.
 * H: m x n row-major bytes; bp_err, out: n bytes; log_probabs: n doubles; syndrome: m bytes (0/1). */
void osd_oracle_postprocess(const uint8_t *H, int64_t m, int64_t n, const uint8_t *syndrome,
                            const uint8_t *bp_err, const double *log_probabs, int64_t osd_order,
                            uint8_t *out)
{
    double *key = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    for (int64_t j = 0; j < n; ++j) {
        double p = exp(log_probabs[j]);                                     /* :53 */
        double q = 1 - p;
        key[j] = (p > q || q != q) ? p : q;                                 /* max.(p, 1 .- p) :55 */
        perm[j] = j;
    }
    merge_sort_desc(key, perm, tmp, 0, n);                                  /* sortperm(rev=true) :55 */
    uint8_t *Hs = (uint8_t *)malloc((size_t)((m > 0 && n > 0) ? m * n : 1));
    uint8_t *es = (uint8_t *)malloc((size_t)(n ? n : 1));
    uint8_t *os = (uint8_t *)malloc((size_t)(n ? n : 1));
    for (int64_t j = 0; j < n; ++j) {                                       /* :56-57 */
        for (int64_t i = 0; i < m; ++i) Hs[i * n + j] = H[i * n + perm[j]];
        es[j] = bp_err[perm[j]];
    }
    if (osd_order == 0) osd0(Hs, m, n, syndrome, es, os);                   /* :59 */
    else osdw(Hs, m, n, syndrome, es, osd_order, os);
    for (int64_t j = 0; j < n; ++j) out[perm[j]] = os[j];                   /* err[invperm(perm)] :60 */
    free(key); free(perm); free(tmp); free(Hs); free(es); free(os);
}
