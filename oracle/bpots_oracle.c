/*
 * bpots_oracle.c -- CPU ORACLE for the BP-OTS decoder (test infrastructure, NOT product code).
 *
 * Literal restatement of src/decoders/bpots_decoder.jl (BPOTSDecoder, reset!, decode!,
 * update_variable_to_check!, update_check_to_variable!, compute_beliefs!), statement order and
 * summation / multiplication order included.  Messages live on the structural non-zeros
 * (CSC edge order); the reference's dense s x n matrices are only ever touched there.
 *
 * PARITY UNPINNED (no Julia runtime, no golden vectors; the reference's tests only assert that
 * the returned estimate reproduces the syndrome).  One deliberate substitution: tanh / atanh
 * are the portable implementations of ldpcdecoders.jl_amd/csrc/portable_math.h (a few ulp from
 * libm, see that header for why) -- the same header the HIP kernel includes, so CPU and GPU take
 * identical data-dependent decisions.  Everything else shares no code with the product.
 * The prior log((1-2p/3)/(2p/3)) (:231) uses libm's log on both sides (host code in both).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Two builds of this one file:
 *   libbpots_oracle.so       tanh / atanh = the portable implementations the HIP kernel uses too: CPU and GPU take
 *                            identical data-dependent decisions, so the GPU tests can demand bit-equality;
 *   libbpots_oracle_libm.so  (-DBPOTS_ORACLE_LIBM) tanh / atanh = the host's libm, sharing NOTHING with the product:
 *                            the independent witness.  Julia's own tanh / atanh are a third correctly rounded-to-~1-ulp
 *                            pair; tests/test_bpots_oracle.py measures how often the two builds reach different
 *                            estimates on the reference's test codes -- the size of the band any such pair lives in. */
#ifdef BPOTS_ORACLE_LIBM
#define pm_tanh tanh
#define pm_atanh atanh
#else
#include "../ldpcdecoders.jl_amd/csrc/portable_math.h"
#endif

typedef struct {
    int64_t s, n, nnz, max_iters, T;
    double per, C;
    int64_t *colptr, *rowval;      /* var_neighbors[j] = rowval[colptr[j]..colptr[j+1])   (:101-109) */
    int64_t *rowptr, *colidx, *pos; /* check_neighbors[i] ascending; pos = CSC edge index of each     */
    double *vc, *cv;               /* messages_vc / messages_cv on the edges (CSC order)            */
    int64_t *osc, *prior_dec, *dec, *best;
    double *prior_llr, *llr, *Pi, *Om;
    int64_t last_iters;
} bpots_oracle;

void bpots_oracle_destroy(bpots_oracle *d)
{
    if (!d) return;
    free(d->colptr); free(d->rowval); free(d->rowptr); free(d->colidx); free(d->pos);
    free(d->vc); free(d->cv); free(d->osc); free(d->prior_dec); free(d->dec); free(d->best);
    free(d->prior_llr); free(d->llr); free(d->Pi); free(d->Om); free(d);
}

bpots_oracle *bpots_oracle_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr,
                                  const int64_t *rowval, double per, int64_t max_iters, int64_t T, double C)
{
    bpots_oracle *d = (bpots_oracle *)calloc(1, sizeof *d);
    d->s = s; d->n = n; d->nnz = nnz; d->per = per; d->max_iters = max_iters; d->T = T; d->C = C;
    size_t e = (size_t)(nnz ? nnz : 1), nn = (size_t)(n ? n : 1);
    d->colptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    d->rowval = (int64_t *)malloc(sizeof(int64_t) * e);
    memcpy(d->colptr, colptr, sizeof(int64_t) * (size_t)(n + 1));
    memcpy(d->rowval, rowval, sizeof(int64_t) * (size_t)nnz);
    d->rowptr = (int64_t *)calloc((size_t)(s + 1), sizeof(int64_t));
    d->colidx = (int64_t *)malloc(sizeof(int64_t) * e);
    d->pos = (int64_t *)malloc(sizeof(int64_t) * e);
    for (int64_t k = 0; k < nnz; ++k) d->rowptr[rowval[k] + 1]++;
    for (int64_t i = 0; i < s; ++i) d->rowptr[i + 1] += d->rowptr[i];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s ? s : 1));
    for (int64_t i = 0; i < s; ++i) fill[i] = d->rowptr[i];
    for (int64_t j = 0; j < n; ++j)                         /* push!(check_neighbors[i], j), j ascending */
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            int64_t q = fill[rowval[k]]++;
            d->colidx[q] = j;
            d->pos[q] = k;
        }
    free(fill);
    d->vc = (double *)calloc(e, sizeof(double));
    d->cv = (double *)calloc(e, sizeof(double));
    d->osc = (int64_t *)calloc(nn, sizeof(int64_t));
    d->prior_dec = (int64_t *)calloc(nn, sizeof(int64_t));
    d->dec = (int64_t *)calloc(nn, sizeof(int64_t));
    d->best = (int64_t *)calloc(nn, sizeof(int64_t));
    d->prior_llr = (double *)calloc(nn, sizeof(double));
    d->llr = (double *)calloc(nn, sizeof(double));
    d->Pi = (double *)calloc(nn, sizeof(double));
    d->Om = (double *)calloc(nn, sizeof(double));
    return d;
}

/* decode!(decoder::BPOTSDecoder, syndrome)  :225-340.  Returns converged; estimate in d->best. */
int bpots_oracle_decode(bpots_oracle *d, const uint8_t *syndrome)
{
    const int64_t s = d->s, n = d->n;
    /* reset!  :142-154 */
    for (int64_t j = 0; j < n; ++j) { d->osc[j] = 0; d->prior_dec[j] = 0; d->prior_llr[j] = 0.0; }
    for (int64_t k = 0; k < d->nnz; ++k) { d->vc[k] = 0.0; d->cv[k] = 0.0; }
    const double prior = log((1 - (2 * d->per / 3)) / (2 * d->per / 3));                 /* :231 */
    for (int64_t j = 0; j < n; ++j) { d->Pi[j] = prior; d->Om[j] = prior; }             /* :231-232 */
    for (int64_t j = 0; j < n; ++j) d->best[j] = 0;                                      /* :235 */
    int64_t best_mismatch = s, best_weight = n;                                          /* :236-237 */
    const double MAX_TANH = 0.99999, MAX_MSG = 100.0;
    d->last_iters = 0;
    for (int64_t iter = 1; iter <= d->max_iters; ++iter) {                               /* :239 */
        d->last_iters = iter;
        for (int64_t j = 0; j < n; ++j)                                                  /* :241-245 */
            for (int64_t k = d->colptr[j]; k < d->colptr[j + 1]; ++k) {
                double msg_sum = 0.0;                                                    /* :161-167 */
                for (int64_t q = d->colptr[j]; q < d->colptr[j + 1]; ++q)
                    if (q != k) msg_sum += d->cv[q];
                d->vc[k] = d->Om[j] + msg_sum;                                           /* :170-171 */
            }
        for (int64_t i = 0; i < s; ++i)                                                  /* :247-251 */
            for (int64_t a = d->rowptr[i]; a < d->rowptr[i + 1]; ++a) {
                double prod_tanh = 1.0;                                                  /* :182-192 */
                for (int64_t b = d->rowptr[i]; b < d->rowptr[i + 1]; ++b)
                    if (b != a) {
                        double t = pm_tanh(0.5 * d->vc[d->pos[b]]);
                        t = fmin(MAX_TANH, fmax(-MAX_TANH, t));
                        prod_tanh *= t;
                    }
                if (syndrome[i] != 0) prod_tanh = -prod_tanh;                            /* :195-197 */
                if (fabs(prod_tanh) >= MAX_TANH) prod_tanh = prod_tanh > 0 ? MAX_TANH : -MAX_TANH;  /* :200-202 */
                double msg = 2.0 * pm_atanh(prod_tanh);                                  /* :203 */
                msg = fmin(MAX_MSG, fmax(-MAX_MSG, msg));                                /* :206-207 */
                d->cv[d->pos[a]] = msg;                                                  /* :209 */
            }
        for (int64_t j = 0; j < n; ++j) {                                                /* compute_beliefs! :120-136 */
            double llr = d->Om[j];
            for (int64_t k = d->colptr[j]; k < d->colptr[j + 1]; ++k) llr += d->cv[k];
            d->llr[j] = llr;
            d->dec[j] = llr < 0.0 ? 1 : 0;
        }
        if (iter > 1)                                                                    /* :257-261 */
            for (int64_t j = 0; j < n; ++j) d->osc[j] += (d->dec[j] ^ d->prior_dec[j]);
        for (int64_t j = 0; j < n; ++j) { d->prior_dec[j] = d->dec[j]; d->prior_llr[j] = d->llr[j]; }  /* :262-263 */
        int64_t mismatch = 0;                                                            /* :266-279 */
        for (int64_t i = 0; i < s; ++i) {
            int64_t acc = 0;
            for (int64_t a = d->rowptr[i]; a < d->rowptr[i + 1]; ++a) acc += d->dec[d->colidx[a]];
            if ((acc % 2) != (int64_t)syndrome[i]) ++mismatch;
        }
        int64_t weight = 0;                                                              /* :281 */
        for (int64_t j = 0; j < n; ++j) weight += d->dec[j];
        if (mismatch < best_mismatch || (mismatch == best_mismatch && weight < best_weight)) {   /* :284-292 */
            best_mismatch = mismatch;
            best_weight = weight;
            for (int64_t j = 0; j < n; ++j) d->best[j] = d->dec[j];
            if (mismatch == 0) return 1;
        }
        if (mismatch > 0 && iter % d->T == 0) {                                          /* :295 */
            for (int64_t j = 0; j < n; ++j) d->Om[j] = d->Pi[j];                         /* :297 */
            int64_t mx = 0;
            for (int64_t j = 0; j < n; ++j) if (d->osc[j] > mx) mx = d->osc[j];
            if (mx > 0) {                                                                /* :300 */
                int64_t max_osc = 0, j1 = -1;
                double min_llr = INFINITY;
                for (int64_t j = 0; j < n; ++j) {                                        /* :305-315 */
                    if (d->osc[j] > max_osc) {
                        max_osc = d->osc[j]; j1 = j; min_llr = fabs(d->llr[j]);
                    } else if (d->osc[j] == max_osc && fabs(d->llr[j]) < min_llr) {
                        j1 = j; min_llr = fabs(d->llr[j]);
                    }
                }
                if (j1 >= 0) { d->osc[j1] = 0; d->Om[j1] = -d->C; }                      /* :318-323 */
                int64_t j2 = 0;                                                          /* :326-333 */
                min_llr = fabs(d->llr[0]);
                for (int64_t j = 1; j < n; ++j)
                    if (fabs(d->llr[j]) < min_llr) { j2 = j; min_llr = fabs(d->llr[j]); }
                d->Om[j2] = -d->C;                                                       /* :336 */
            }
        }
    }
    return 0;                                                                            /* :340 */
}

void bpots_oracle_decode_batch(bpots_oracle *d, int64_t B, const uint8_t *syndromes, uint8_t *errors,
                               uint8_t *converged, int32_t *iters)
{
    for (int64_t b = 0; b < B; ++b) {
        converged[b] = (uint8_t)bpots_oracle_decode(d, syndromes + (size_t)b * (size_t)d->s);
        for (int64_t j = 0; j < d->n; ++j) errors[(size_t)b * (size_t)d->n + (size_t)j] = (uint8_t)d->best[j];
        if (iters) iters[b] = (int32_t)d->last_iters;
    }
}

#ifndef BPOTS_ORACLE_LIBM
/* the portable functions, exported for tests (accuracy vs libm; the Python restatement uses them too) */
double pm_tanh_export(double x) { return pm_tanh(x); }
double pm_atanh_export(double x) { return pm_atanh(x); }
double pm_exp_export(double x) { return pm_exp(x); }
double pm_log_export(double x) { return pm_log(x); }
#endif
