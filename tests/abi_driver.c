/* abi_driver.c -- plain-C consumer of include/ldpc_mi355x.h (compiled by tests/test_abi_cpu.py with
 * gcc -std=c99 and linked against libldpc_mi355x.so): proves the header is valid C and that a
 * non-Python host can drive the library.  Without a GPU it checks the no-device error path; with
 * one (argv[1] = "gpu") it decodes the all-zero syndrome of a tiny code and the OSD host step. */
#include <stdio.h>
#include <string.h>
#include "ldpc_mi355x.h"

int main(int argc, char **argv)
{
    /* H = [1 1 0 ; 0 1 1] as zero-based CSC */
    const int64_t colptr[4] = {0, 1, 3, 4};
    const int64_t rowval[4] = {0, 0, 1, 1};
    ldpc_bp_decoder *dec = NULL;
    ldpc_osd *osd = NULL;
    if (ldpc_abi_version() != LDPC_MI355X_ABI_VERSION) return 10;
    if (strcmp(ldpc_build_target(), "gfx950") != 0) return 11;
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
