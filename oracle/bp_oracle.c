/*
 * bp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the belief-propagation hot path of
 * QuantumSavory/LDPCDecoders.jl, operation for operation, in IEEE-754 double:
 *
 *   scratch init   src/decoders/belief_propagation.jl:20-22
 *   constructor    src/decoders/belief_propagation.jl:61-67   (CSC of H and of H')
 *   reset!         src/decoders/belief_propagation.jl:83-91
 *   decode!        src/decoders/belief_propagation.jl:121-188
 *   batchdecode!   src/decoders/belief_propagation.jl:220-231,
 *                  src/decoders/abstract_decoder.jl:31-48
 *
 * PARITY UNPINNED: the reference is Julia, no Julia runtime exists in this
 * image, and the reference's own tests hold no golden vectors.
 * This restatement is therefore pinned only by (1) an independently written
 * pure-Python restatement (oracle/bp_reference_py.py) that must agree bit for
 * bit, and (2) the statistical acceptance tests of test/test_bp_decoder.jl.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  The product (libldpc_mi355x.so) never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  No
 * FMA contraction and no re-association: Julia performs neither.
 *
 * Two storage modes with identical arithmetic:
 *   dense = 1 : the reference's own cost structure -- two dense s*n Float64
 *               column-major matrices, fully zero-filled by reset! on every
 *               decode (belief_propagation.jl:87-88) and addressed [i + j*s].
 *   dense = 0 : messages kept on the structural non-zeros only (edge list in
 *               CSC order).  Value-identical because the reference only ever
 *               touches structural non-zeros.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t s, n, nnz;
    int64_t max_iters;
    double per;
    int dense;
    /* sparse_H: CSC of H (columns = bits), 0-based here. belief_propagation.jl:63 */
    int64_t *colptr;  /* n+1 */
    int64_t *rowval;  /* nnz : check index of each edge, ascending inside a column */
    /* sparse_HT: CSC of H' (columns = checks). belief_propagation.jl:64 */
    int64_t *colptrT; /* s+1 */
    int64_t *rowvalT; /* nnz : bit index of each edge, ascending inside a check */
    int64_t *posT;    /* nnz : for edge k of sparse_HT, its edge index in sparse_H */
    /* scratch. belief_propagation.jl:3-22 */
    double *log_probabs;   /* n */
    double *channel_probs; /* n */
    double *bit_2_check;   /* dense: s*n, else nnz (CSC edge order) */
    double *check_2_bit;   /* same */
    double *err;           /* n */
    int64_t last_iters;
} bp_oracle;

void bp_oracle_destroy(bp_oracle *d)
{
    if (!d) return;
    free(d->colptr); free(d->rowval); free(d->colptrT); free(d->rowvalT); free(d->posT);
    free(d->log_probabs); free(d->channel_probs); free(d->bit_2_check); free(d->check_2_bit);
    free(d->err); free(d);
}

/* BeliefPropagationDecoder(H, per, max_iters): belief_propagation.jl:61-67.
 * H is handed over as the CSC pattern `sparse(H)` would produce (0-based). */
bp_oracle *bp_oracle_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr,
                            const int64_t *rowval, double per, int64_t max_iters, int dense)
{
    if (s < 0 || n < 0 || nnz < 0 || colptr[0] != 0 || colptr[n] != nnz) return NULL;
    for (int64_t j = 0; j < n; ++j) {
        if (colptr[j + 1] < colptr[j]) return NULL;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (rowval[k] < 0 || rowval[k] >= s) return NULL;
            if (k > colptr[j] && rowval[k] <= rowval[k - 1]) return NULL;
        }
    }
    bp_oracle *d = (bp_oracle *)calloc(1, sizeof *d);
    d->s = s; d->n = n; d->nnz = nnz; d->per = per; d->max_iters = max_iters; d->dense = dense;
    d->colptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    d->rowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    memcpy(d->colptr, colptr, sizeof(int64_t) * (size_t)(n + 1));
    memcpy(d->rowval, rowval, sizeof(int64_t) * (size_t)nnz);
    /* sparse(H'): bits of every check in ascending order. */
    d->colptrT = (int64_t *)calloc((size_t)(s + 1), sizeof(int64_t));
    d->rowvalT = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    d->posT = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz ? nnz : 1));
    for (int64_t k = 0; k < nnz; ++k) d->colptrT[rowval[k] + 1]++;
    for (int64_t i = 0; i < s; ++i) d->colptrT[i + 1] += d->colptrT[i];
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s ? s : 1));
    for (int64_t i = 0; i < s; ++i) fill[i] = d->colptrT[i];
    for (int64_t j = 0; j < n; ++j)
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            int64_t q = fill[rowval[k]]++;
            d->rowvalT[q] = j;
            d->posT[q] = k;
        }
    free(fill);
    size_t msz = dense ? (size_t)s * (size_t)n : (size_t)nnz;
    if (msz == 0) msz = 1;
    d->log_probabs = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    d->channel_probs = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    d->err = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    d->bit_2_check = (double *)calloc(msz, sizeof(double));
    d->check_2_bit = (double *)calloc(msz, sizeof(double));
    if (!d->bit_2_check || !d->check_2_bit) { bp_oracle_destroy(d); return NULL; }
    for (int64_t j = 0; j < n; ++j) d->channel_probs[j] = per; /* fill(per, n) :21 */
    return d;
}

/* reset!: belief_propagation.jl:83-91 */
void bp_oracle_reset(bp_oracle *d)
{
    size_t msz = d->dense ? (size_t)d->s * (size_t)d->n : (size_t)d->nnz;
    for (int64_t j = 0; j < d->n; ++j) d->log_probabs[j] = 0.0;
    for (int64_t j = 0; j < d->n; ++j) d->channel_probs[j] = d->per;
    memset(d->bit_2_check, 0, msz * sizeof(double));
    memset(d->check_2_bit, 0, msz * sizeof(double));
    for (int64_t j = 0; j < d->n; ++j) d->err[j] = 0.0;
}

/* (-1)^x for an integer x, as a Float64 (belief_propagation.jl:136). */
static inline double minus_one_pow(uint8_t x) { return (x & 1u) ? -1.0 : 1.0; }

/* decode!: belief_propagation.jl:121-188.  Syndrome entries are uint8; an entry
 * other than 0/1 keeps its parity for the sign (like (-1)^x) and can never
 * satisfy the `==` of the convergence test (:181), exactly as an Int 2 or 3
 * would in the reference.  Returns converged (0/1); result lives in d->err,
 * d->log_probabs (aliases, like :187). */
int bp_oracle_decode(bp_oracle *d, const uint8_t *syndrome)
{
    const int64_t s = d->s, n = d->n;
    const int dense = d->dense;
    double *b2c = d->bit_2_check, *c2b = d->check_2_bit;
    bp_oracle_reset(d);                                                    /* :122 */
#define B2C(i, j, e) b2c[dense ? ((size_t)(i) + (size_t)(j) * (size_t)s) : (size_t)(e)]
#define C2B(i, j, e) c2b[dense ? ((size_t)(i) + (size_t)(j) * (size_t)s) : (size_t)(e)]
    for (int64_t j = 0; j < n; ++j)                                        /* :127-131 */
        for (int64_t k = d->colptr[j]; k < d->colptr[j + 1]; ++k)
            B2C(d->rowval[k], j, k) = d->channel_probs[j] / (1 - d->channel_probs[j]);

    int converged = 0;
    d->last_iters = 0;
    for (int64_t iter = 1; iter <= d->max_iters; ++iter) {                 /* :134 */
        d->last_iters = iter;
        for (int64_t i = 0; i < s; ++i) {                                  /* :135-150 */
            double temp = minus_one_pow(syndrome[i]);                      /* :136 */
            for (int64_t k = d->colptrT[i]; k < d->colptrT[i + 1]; ++k) {  /* :137-141 */
                int64_t j = d->rowvalT[k], e = d->posT[k];
                C2B(i, j, e) = temp;
                temp *= 2 / (1 + B2C(i, j, e)) - 1;
            }
            temp = 1.0;                                                    /* :143 */
            for (int64_t k = d->colptrT[i + 1] - 1; k >= d->colptrT[i]; --k) { /* :144-149 */
                int64_t j = d->rowvalT[k], e = d->posT[k];
                C2B(i, j, e) *= temp;
                C2B(i, j, e) = (1 - C2B(i, j, e)) / (1 + C2B(i, j, e));
                temp *= 2 / (1 + B2C(i, j, e)) - 1;
            }
        }
        for (int64_t j = 0; j < n; ++j) {                                  /* :152-178 */
            double temp = d->channel_probs[j] / (1 - d->channel_probs[j]); /* :153 */
            for (int64_t k = d->colptr[j]; k < d->colptr[j + 1]; ++k) {    /* :155-161 */
                int64_t i = d->rowval[k];
                B2C(i, j, k) = temp;
                temp *= C2B(i, j, k);
                if (isnan(temp)) temp = 1.0;
            }
            d->log_probabs[j] = log(1 / temp);                             /* :163 */
            if (temp >= 1) d->err[j] = 1; else d->err[j] = 0;              /* :164-168 */
            temp = 1.0;                                                    /* :170 */
            for (int64_t k = d->colptr[j + 1] - 1; k >= d->colptr[j]; --k) { /* :171-177 */
                int64_t i = d->rowval[k];
                B2C(i, j, k) *= temp;
                temp *= C2B(i, j, k);
                if (isnan(temp)) temp = 1.0;
            }
        }
        /* syndrome_decoded = (sparse_H * err) .% 2 ; all(.== syndrome)  :180-184 */
        int all_eq = 1;
        for (int64_t i = 0; i < s && all_eq; ++i) {
            double acc = 0.0;
            for (int64_t k = d->colptrT[i]; k < d->colptrT[i + 1]; ++k) acc += d->err[d->rowvalT[k]];
            double dec = fmod(acc, 2.0);
            if (dec != (double)syndrome[i]) all_eq = 0;
        }
        if (all_eq) { converged = 1; break; }
    }
#undef B2C
#undef C2B
    return converged;                                                      /* :187 */
}

/* batchdecode!: belief_propagation.jl:220-231.  syndromes is [B][s] (the
 * columns of Julia's s x B matrix, each contiguous), errors is [B][n].
 * llr / iters may be NULL. */
void bp_oracle_decode_batch(bp_oracle *d, int64_t B, const uint8_t *syndromes, uint8_t *errors,
                            uint8_t *converged, double *llr, int32_t *iters)
{
    for (int64_t b = 0; b < B; ++b) {                                      /* :224 */
        int conv = bp_oracle_decode(d, syndromes + (size_t)b * (size_t)d->s);
        converged[b] = (uint8_t)conv;                                      /* :226 */
        for (int64_t j = 0; j < d->n; ++j)                                 /* :227 */
            errors[(size_t)b * (size_t)d->n + (size_t)j] = (uint8_t)(d->err[j] != 0.0);
        if (llr) memcpy(llr + (size_t)b * (size_t)d->n, d->log_probabs, sizeof(double) * (size_t)d->n);
        if (iters) iters[b] = (int32_t)d->last_iters;
    }
}

/* Accessors for the cross-check against the Python restatement. */
const double *bp_oracle_err(const bp_oracle *d) { return d->err; }
const double *bp_oracle_log_probabs(const bp_oracle *d) { return d->log_probabs; }
int64_t bp_oracle_last_iters(const bp_oracle *d) { return d->last_iters; }
/* Copies the messages at the structural non-zeros, CSC edge order, whichever
 * storage mode is in use. */
void bp_oracle_messages(const bp_oracle *d, double *b2c_out, double *c2b_out)
{
    for (int64_t j = 0; j < d->n; ++j)
        for (int64_t k = d->colptr[j]; k < d->colptr[j + 1]; ++k) {
            size_t at = d->dense ? ((size_t)d->rowval[k] + (size_t)j * (size_t)d->s) : (size_t)k;
            b2c_out[k] = d->bit_2_check[at];
            c2b_out[k] = d->check_2_bit[at];
        }
}
