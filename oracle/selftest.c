/* Sanitizer self-test of the CPU oracle (test infrastructure): built with
 * -fsanitize=address,undefined by `make selftest` and run by tests/test_oracle_golden.py.
 * Exercises every structure family at small sizes through both entry points. */
#include "wofdm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static int run_case(int n, int k, int S, int cp, int cs, int ttx, int trx, int rm, int shift, int taps, int matlab)
{
    wofdm_oracle_sys sys = {n, k, S, cp, cs, ttx, trx, rm, shift, taps, matlab};
    int P = n + cp + cs, NW = n + trx, i, rc;
    double *wtx = malloc(sizeof(double) * P), *wrx = malloc(sizeof(double) * NW);
    double *h = calloc(2 * (size_t)taps, sizeof(double));
    double snr[2] = {5.0, 25.0};
    uint64_t counts[2 * 4] = {0};
    for (i = 0; i < P; i++) wtx[i] = 1.0;
    for (i = 0; i < NW; i++) wrx[i] = 1.0;
    for (i = 0; i < taps; i++) { h[2 * i] = 1.0 / (1 + i); h[2 * i + 1] = 0.1 * i; }
    rc = wofdm_oracle_run(&sys, 1, 2, 1, wtx, wrx, h, snr, 7, 3, 5, 2, counts);
    if (rc == 0 && (counts[1] != (uint64_t)5 * (S - 1) * n * k || counts[0] < counts[4])) rc = -1000;
    free(wtx); free(wrx); free(h);
    if (rc) fprintf(stderr, "case N=%d k=%d S=%d failed: %d\n", n, k, S, rc);
    return rc;
}

int main(void)
{
    int bad = 0;
    bad |= run_case(64, 2, 16, 16, 8, 8, 0, 16, 0, 21, 1);     /* wtx   */
    bad |= run_case(64, 4, 16, 16, 5, 0, 10, 11, 0, 21, 0);    /* wrx   */
    bad |= run_case(64, 6, 7, 16, 8, 8, 10, 6, 5, 21, 1);      /* WOLA  */
    bad |= run_case(128, 4, 2, 20, 13, 8, 10, 15, 0, 5, 1);    /* CPW   */
    bad |= run_case(64, 2, 3, 16, 0, 8, 0, 8, 8, 1, 1);        /* CPwtx, one tap */
    bad |= run_case(256, 4, 16, 32, 0, 0, 10, 22, 5, 21, 0);   /* CPwrx */
    {   /* invalid geometry must be rejected, not read out of bounds */
        wofdm_oracle_sys sys = {64, 2, 16, 16, 8, 8, 0, 3, 0, 21, 1};
        double w[200] = {0}, h[42] = {0}, snr = 1.0;
        uint64_t c[4] = {0};
        if (wofdm_oracle_run(&sys, 1, 1, 1, w, w, h, &snr, 1, 0, 1, 1, c) == 0) bad = 1;
    }
    printf(bad ? "selftest FAILED\n" : "selftest ok\n");
    return bad != 0;
}
