/* oracle_sanitize.c -- the CPU oracles (oracle/bp_oracle.c in both storage modes, oracle/bpots_oracle.c)
 * under AddressSanitizer + UndefinedBehaviorSanitizer on random sparse graphs (irregular, with empty
 * rows / columns), channel probabilities from 0 to 1 and non-binary syndrome entries.  The checker
 * itself must not read out of bounds: a golden vector made by an oracle with such a bug pins nothing.
 * Also asserts that the edge-list and the reference-faithful dense storage modes agree bit for bit.
 * Built and run by tests/test_oracle_cross.py::test_oracles_under_sanitizers. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct bp_oracle bp_oracle;
bp_oracle *bp_oracle_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                            double per, int64_t max_iters, int dense);
void bp_oracle_destroy(bp_oracle *d);
void bp_oracle_decode_batch(bp_oracle *d, int64_t B, const uint8_t *syndromes, uint8_t *errors, uint8_t *converged,
                            double *llr, int32_t *iters);
typedef struct bpots_oracle bpots_oracle;
bpots_oracle *bpots_oracle_create(int64_t s, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                                  double per, int64_t max_iters, int64_t T, double C);
void bpots_oracle_destroy(bpots_oracle *d);
void bpots_oracle_decode_batch(bpots_oracle *d, int64_t B, const uint8_t *syndromes, uint8_t *errors,
                               uint8_t *converged, int32_t *iters);

static uint64_t st = 0x2545F4914F6CDD1Dull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static double uni(void) { return (double)(rnd() >> 11) / 9007199254740992.0; }

int main(void)
{
    const double pers[] = {0.0, 1e-9, 0.01, 0.05, 0.3, 0.5, 0.97, 1.0};
    int cases = 0;
    for (int trial = 0; trial < 48; ++trial) {
        const int64_t s = (trial == 0) ? 0 : 1 + (int64_t)(rnd() % 30), n = 1 + (int64_t)(rnd() % 60);
        const double dens = 0.04 + 0.25 * uni();
        int64_t *colptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
        int64_t *rowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s * n + 1));
        int64_t nnz = 0;
        for (int64_t j = 0; j < n; ++j) {
            for (int64_t i = 0; i < s; ++i)
                if (j != 3 && i != 2 && uni() < dens) rowval[nnz++] = i;      /* column 3 and row 2 stay empty */
            colptr[j + 1] = nnz;
        }
        const int64_t B = 1 + (int64_t)(rnd() % 9);
        uint8_t *syn = (uint8_t *)malloc((size_t)(B * s + 1));
        for (int64_t k = 0; k < B * s; ++k) syn[k] = (uint8_t)((rnd() % 23 == 0) ? 2 + rnd() % 3 : rnd() & 1);
        const double per = pers[trial % 8];
        const int64_t iters = (trial % 5 == 4) ? 0 : 1 + (int64_t)(rnd() % 25);
        uint8_t *e[2], *c[2];
        double *l[2];
        int32_t *it[2];
        for (int mode = 0; mode < 2; ++mode) {
            e[mode] = (uint8_t *)malloc((size_t)(B * n)); c[mode] = (uint8_t *)malloc((size_t)B);
            l[mode] = (double *)malloc(sizeof(double) * (size_t)(B * n)); it[mode] = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
            bp_oracle *d = bp_oracle_create(s, n, nnz, colptr, rowval, per, iters, mode);
            if (!d) { printf("bp_oracle_create failed\n"); return 2; }
            bp_oracle_decode_batch(d, B, syn, e[mode], c[mode], l[mode], it[mode]);
            bp_oracle_destroy(d);
        }
        if (memcmp(e[0], e[1], (size_t)(B * n)) || memcmp(c[0], c[1], (size_t)B) || memcmp(it[0], it[1], sizeof(int32_t) * (size_t)B) ||
            memcmp(l[0], l[1], sizeof(double) * (size_t)(B * n))) {
            printf("MISMATCH between storage modes, trial %d\n", trial);
            return 1;
        }
        if (per > 0.0 && per < 1.0) {   /* BP-OTS takes log((1-2p/3)/(2p/3)) */
            bpots_oracle *o = bpots_oracle_create(s, n, nnz, colptr, rowval, per, iters, 1 + (int64_t)(rnd() % 9), 1.0 + 2.0 * uni());
            if (!o) { printf("bpots_oracle_create failed\n"); return 2; }
            for (int64_t k = 0; k < B * s; ++k) syn[k] &= 1u;
            bpots_oracle_decode_batch(o, B, syn, e[0], c[0], it[0]);
            bpots_oracle_destroy(o);
        }
        for (int mode = 0; mode < 2; ++mode) { free(e[mode]); free(c[mode]); free(l[mode]); free(it[mode]); }
        free(colptr); free(rowval); free(syn);
        cases += (int)B;
    }
    printf("OK %d syndromes, storage modes identical\n", cases);
    return 0;
}
