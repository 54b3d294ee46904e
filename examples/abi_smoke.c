/* Plain-C caller of the C ABI (include/pyrad_hip.h): what a non-Python host would write.
 * One CO2-like line on the C1 grid, host in / host out (lbl_xsec_accumulate), then the peak of the
 * cross section against the closed-form pseudo-Voigt value of SURVEY.md §8c.
 *   gcc -std=c99 -Iinclude examples/abi_smoke.c -Lpyrad_amd/lib -lpyrad_hip -Wl,-rpath,$PWD/pyrad_amd/lib -o abi_smoke
 * tests/test_abi_cpu.py compiles it (the header must stay valid C99) and, on a GPU box, runs it. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pyrad_hip.h"

int main(void) {
    lbl_ctx* ctx = NULL;
    int rc = lbl_ctx_create(0, &ctx);
    if (rc != LBL_OK) {
        fprintf(stderr, "lbl_ctx_create: %d (%s)\n", rc, lbl_last_error(NULL));
        return rc == LBL_ERR_NO_DEVICE ? 77 : 1;      /* 77: no GPU here - nothing to run, nothing faked */
    }
    /* pyradClasses.py:648-676 for Layer(10, 296, 1013.25, 600, 700): resolution 0.01, window 500 points */
    lbl_grid grid = {600.0, 700.0, 0.01, 0.01, 10000, 10000, 500, 0, 0};
    lbl_iso_params iso = {296.0, 1013.25, 400e-6, 43.98983, 286.09, 286.09};
    double nu = 650.003, sw = 1e-20, elower = 1000.0, g_air = 0.07, g_self = 0.09, n_air = 0.7, d_air = -0.002;
    double* xsec = (double*)malloc(sizeof(double) * (size_t)grid.n_base);
    int64_t counts[3] = {0, 0, 0};
    rc = lbl_xsec_accumulate(ctx, &nu, &sw, &elower, &g_air, &g_self, &n_air, &d_air, 1, &iso, &grid, xsec, counts);
    if (rc != LBL_OK) {
        fprintf(stderr, "lbl_xsec_accumulate: %d (%s)\n", rc, lbl_last_error(ctx));
        return 1;
    }
    /* the reference's value at the line centre (index 5000), measured in SURVEY.md §8c */
    const double expect = 4.5462648814858876e-20;
    const double err = fabs(xsec[5000] - expect) / expect;
    printf("peak %.17g (reference %.17g, rel err %.2e); regimes gaussian/lorentz/voigt = %lld/%lld/%lld; support [%d..%d]\n",
           xsec[5000], expect, err, (long long)counts[0], (long long)counts[1], (long long)counts[2],
           xsec[4501] == 0.0 ? 4502 : -1, xsec[5499] == 0.0 ? 5498 : -1);
    free(xsec);
    lbl_ctx_destroy(ctx);
    return err <= 1e-12 ? 0 : 2;
}
