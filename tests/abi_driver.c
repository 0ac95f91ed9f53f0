/* abi_driver.c -- plain-C consumer of include/ldpc_mi355x.h (compiled by tests/test_abi_cpu.py with
 * gcc -std=c99 and linked against libldpc_mi355x.so): proves the header is valid C and that a
 * non-Python host can drive the library.  Without a GPU it checks the no-device error path; with
 * one (argv[1] = "gpu") it decodes the all-zero syndrome of a tiny code and the OSD host step;
 * argv[1] = "multi": the multi-device entries (ldpc_bp_create_multi ...) against the single-device one. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ldpc_mi355x.h"
#include "ldpc_mi355x_debug.h" /* (must be valid C too) */

/* the few HIP runtime entries a C host needs for device buffers (libamdhip64 is already a dependency of the library) */
extern int hipMalloc(void **p, size_t bytes);
extern int hipFree(void *p);
extern int hipMemcpy(void *dst, const void *src, size_t bytes, int kind); /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);
extern int hipMemset(void *p, int v, size_t bytes);

#define MS 48
#define MN 96
#define MB 1000

/* one multi-device configuration against the reference outputs of the single-device entry; 0 = equal */
static int run_multi(int ndev, int exchange, int64_t nnz, const int64_t *colptr, const int64_t *rowval, const uint8_t *syn,
                     const uint8_t *d_syn, const uint8_t *r_err, const uint8_t *r_conv, const int32_t *r_it, const double *r_llr)
{
    const int32_t devs[2] = {0, 0};
    ldpc_bp_multi *m = NULL;
    ldpc_bp_multi_info mi;
    uint8_t *err = malloc((size_t)MB * MN), *conv = malloc(MB), *d_err = NULL, *d_conv = NULL;
    int32_t *it = malloc(sizeof(int32_t) * MB), *d_it = NULL;
    double *llr = malloc(sizeof(double) * MB * MN), *d_llr = NULL;
    int rc = 0;
    if (ldpc_bp_create_multi(ndev, devs, exchange, MS, MN, nnz, colptr, rowval, 0.04, 20, NULL, &m) != LDPC_OK) {
        fprintf(stderr, "create_multi(%d, %d): %s\n", ndev, exchange, ldpc_last_error());
        return 40;
    }
    if (!ldpc_bp_multi_handle(m, ndev - 1) || ldpc_bp_multi_handle(m, ndev)) rc = 41;
    /* host form: one call, one caller-held matrix */
    memset(err, 9, (size_t)MB * MN); memset(conv, 9, MB);
    if (!rc && ldpc_bp_decode_batch_multi(m, MB, syn, err, conv, llr, it) != LDPC_OK) { fprintf(stderr, "host form: %s\n", ldpc_last_error()); rc = 42; }
    if (!rc && (memcmp(err, r_err, (size_t)MB * MN) || memcmp(conv, r_conv, MB) || memcmp(it, r_it, sizeof(int32_t) * MB) ||
                memcmp(llr, r_llr, sizeof(double) * MB * MN))) rc = 43;
    /* root-device form */
    if (hipMalloc((void **)&d_err, (size_t)MB * MN) || hipMalloc((void **)&d_conv, MB) || hipMalloc((void **)&d_it, sizeof(int32_t) * MB) ||
        hipMalloc((void **)&d_llr, sizeof(double) * MB * MN)) rc = rc ? rc : 44;
    for (int rep = 0; rep < 2 && !rc; ++rep) { /* twice: the shard buffers and communicators are reused */
        hipMemset(d_err, 9, (size_t)MB * MN); hipMemset(d_conv, 9, MB);
        if (ldpc_bp_decode_batch_multi_device(m, MB, d_syn, d_err, d_conv, d_llr, d_it, NULL) != LDPC_OK) { fprintf(stderr, "device form: %s\n", ldpc_last_error()); rc = 45; break; }
        if (ldpc_bp_multi_last_status(m) != LDPC_OK) { rc = 46; break; }
        hipDeviceSynchronize();
        hipMemcpy(err, d_err, (size_t)MB * MN, 2); hipMemcpy(conv, d_conv, MB, 2);
        hipMemcpy(it, d_it, sizeof(int32_t) * MB, 2); hipMemcpy(llr, d_llr, sizeof(double) * MB * MN, 2);
        if (memcmp(err, r_err, (size_t)MB * MN) || memcmp(conv, r_conv, MB) || memcmp(it, r_it, sizeof(int32_t) * MB) ||
            memcmp(llr, r_llr, sizeof(double) * MB * MN)) rc = 47;
    }
    if (!rc && (ldpc_bp_multi_get_info(m, &mi) != LDPC_OK || mi.ndev != ndev ||
                mi.exchange != (ndev == 1 ? (exchange == LDPC_EXCHANGE_RCCL ? LDPC_EXCHANGE_RCCL : LDPC_EXCHANGE_NONE) : LDPC_EXCHANGE_COPY)))
        rc = 48;
    if (!rc) printf("  ndev %d exchange %d: host and root-device form equal the single-device entry (scatter %.3f ms, gather %.3f ms)\n",
                    ndev, mi.exchange, mi.scatter_ms, mi.gather_ms);
    hipFree(d_err); hipFree(d_conv); hipFree(d_it); hipFree(d_llr);
    ldpc_bp_destroy_multi(m);
    free(err); free(conv); free(it); free(llr);
    return rc;
}

static int multi_main(void)
{
    /* a small irregular code: bit j sits in checks j % 48, (5j + 7) % 48, (11j + 3) % 48 (duplicates dropped) */
    static int64_t colptr[MN + 1], rowval[3 * MN];
    static uint8_t syn[MB * MS], r_err[MB * MN], r_conv[MB];
    static int32_t r_it[MB];
    static double r_llr[MB * MN];
    int64_t nnz = 0;
    uint32_t lcg = 12345u;
    ldpc_bp_decoder *dec = NULL;
    uint8_t *d_syn = NULL, *d_err = NULL, *d_conv = NULL;
    int32_t *d_it = NULL;
    double *d_llr = NULL;
    int rc;
    for (int j = 0; j < MN; ++j) {
        int64_t r[3] = {j % MS, (5 * j + 7) % MS, (11 * j + 3) % MS};
        colptr[j] = nnz;
        for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (r[b] < r[a]) { int64_t q = r[a]; r[a] = r[b]; r[b] = q; }
        for (int a = 0; a < 3; ++a) if (a == 0 || r[a] != r[a - 1]) rowval[nnz++] = r[a];
    }
    colptr[MN] = nnz;
    /* syndromes of sparse random errors (so that most converge, some do not) */
    memset(syn, 0, sizeof syn);
    for (int b = 0; b < MB; ++b)
        for (int j = 0; j < MN; ++j) {
            lcg = lcg * 1664525u + 1013904223u;
            if ((lcg >> 8) % 100u < (unsigned)(b % 9))
                for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) syn[b * MS + rowval[k]] ^= 1;
        }
    /* reference: the single-device, device-pointer entry */
    if (ldpc_bp_create(MS, MN, nnz, colptr, rowval, 0.04, 20, NULL, &dec) != LDPC_OK) { fprintf(stderr, "create: %s\n", ldpc_last_error()); return 30; }
    if (hipMalloc((void **)&d_syn, sizeof syn) || hipMalloc((void **)&d_err, sizeof r_err) || hipMalloc((void **)&d_conv, sizeof r_conv) ||
        hipMalloc((void **)&d_it, sizeof r_it) || hipMalloc((void **)&d_llr, sizeof r_llr)) return 31;
    hipMemcpy(d_syn, syn, sizeof syn, 1);
    if (ldpc_bp_decode_batch_device(dec, MB, d_syn, d_err, d_conv, d_llr, d_it, NULL) != LDPC_OK || ldpc_bp_last_status(dec) != LDPC_OK) return 32;
    hipDeviceSynchronize();
    hipMemcpy(r_err, d_err, sizeof r_err, 2); hipMemcpy(r_conv, d_conv, sizeof r_conv, 2);
    hipMemcpy(r_it, d_it, sizeof r_it, 2); hipMemcpy(r_llr, d_llr, sizeof r_llr, 2);
    ldpc_bp_destroy(dec);
    {
        int nconv = 0;
        for (int b = 0; b < MB; ++b) nconv += r_conv[b];
        if (nconv < MB / 10 || nconv > MB - MB / 10) { fprintf(stderr, "test batch is not mixed (%d converged)\n", nconv); return 33; }
    }
    if ((rc = run_multi(1, LDPC_EXCHANGE_AUTO, nnz, colptr, rowval, syn, d_syn, r_err, r_conv, r_it, r_llr)) != 0) return rc;
    if ((rc = run_multi(2, LDPC_EXCHANGE_AUTO, nnz, colptr, rowval, syn, d_syn, r_err, r_conv, r_it, r_llr)) != 0) return rc + 100;
    if ((rc = run_multi(1, LDPC_EXCHANGE_RCCL, nnz, colptr, rowval, syn, d_syn, r_err, r_conv, r_it, r_llr)) != 0) return rc + 200;
    hipFree(d_syn); hipFree(d_err); hipFree(d_conv); hipFree(d_it); hipFree(d_llr);
    printf("abi_driver multi ok\n");
    return 0;
}

int main(int argc, char **argv)
{
    /* H = [1 1 0 ; 0 1 1] as zero-based CSC */
    const int64_t colptr[4] = {0, 1, 3, 4};
    const int64_t rowval[4] = {0, 0, 1, 1};
    ldpc_bp_decoder *dec = NULL;
    ldpc_osd *osd = NULL;
    if (ldpc_abi_version() != LDPC_MI355X_ABI_VERSION) return 10;
    if (strcmp(ldpc_build_target(), "gfx950") != 0) return 11;
    if (argc > 1 && strcmp(argv[1], "multi") == 0) return multi_main();
    /* host-only part of the ABI works everywhere */
    if (ldpc_osd_create(2, 3, 4, colptr, rowval, 0, &osd) != LDPC_OK) return 12;
    {
        const uint8_t syn[2] = {1, 0}, bp[3] = {0, 0, 0};
        const double llr[3] = {3.0, 1.0, 2.0};
        uint8_t out[3] = {9, 9, 9};
        if (ldpc_osd_postprocess_batch(osd, 1, syn, bp, llr, out, 1) != LDPC_OK) return 13;
        if (((out[0] ^ out[1]) != 1) || ((out[1] ^ out[2]) != 0)) return 14; /* H*out == syn */
    }
    ldpc_osd_destroy(osd);
    ldpc_status st = ldpc_bp_create(2, 3, 4, colptr, rowval, 0.1, 10, NULL, &dec);
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
        uint8_t syn[2] = {0, 0}, err[3] = {9, 9, 9}, conv = 9;
        double llr[3];
        int32_t it = -1;
        ldpc_bp_info info;
        if (st != LDPC_OK) { fprintf(stderr, "create: %s\n", ldpc_last_error()); return 20; }
        if (ldpc_bp_decode_batch(dec, 1, syn, err, &conv, llr, &it) != LDPC_OK) return 21;
        if (conv != 1 || it != 1 || err[0] || err[1] || err[2]) return 22;
        if (ldpc_bp_get_info(dec, &info) != LDPC_OK || info.n != 3 || info.s != 2) return 23;
        ldpc_bp_destroy(dec);
        printf("abi_driver gpu ok\n");
        return 0;
    }
    if (ldpc_device_count() == 0) {
        if (st != LDPC_ERR_NO_DEVICE || dec != NULL) return 30;
        if (strlen(ldpc_last_error()) == 0) return 31;
    } else if (st == LDPC_OK) {
        ldpc_bp_destroy(dec);
    }
    printf("abi_driver ok\n");
    return 0;
}
