/* tests/capi_c/capi_smoke.c -- the drop-in boundary used from PLAIN C (what an R .Call shim or any C host does):
 * include/ssde.h compiles as C99, libssde_hip.so links without C++ or Python, and one small problem goes through
 * ssde_create / ssde_info / ssde_eval (fn then gr: the memo) / ssde_laplace_eval / ssde_report / ssde_destroy.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include tests/capi_c/capi_smoke.c -L smoothsde_amd/lib -lssde_hip -lm
 *
 * Input: a deterministic 2-D CTCRW batch of `n_tracks` tracks x `rows` rows built here (a linear congruential stream,
 * so that the Python test can rebuild the same arrays), optional spline-like random-effect block on tau.
 * Output: one line "value grad0 grad1 ... | laplace_value | n_evals n_memo_hits" with %.17g numbers, or
 * "create failed: <status> <message>" and exit code 3 when no gfx950 device is visible (the CPU suite expects that). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssde.h"

static double lcg(unsigned long long *s) {           /* uniform in (0, 1) */
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return ((double)((*s >> 11) + 1ULL)) / 9007199254740994.0;
}

int main(int argc, char **argv) {
    const int n_tracks = argc > 1 ? atoi(argv[1]) : 6, rows = argc > 2 ? atoi(argv[2]) : 40, with_re = argc > 3 ? atoi(argv[3]) : 1;
    const long n = (long)n_tracks * rows;
    const int d = 2, q = 4, K = with_re ? 3 : 0;
    double *id = malloc(sizeof(double) * n), *times = malloc(sizeof(double) * n), *obs = malloc(sizeof(double) * n * d);
    double *xre = malloc(sizeof(double) * n * (K > 0 ? K : 1));
    unsigned long long seed = 12345ULL;
    for (int m = 0; m < n_tracks; m++) {
        double x = 0.0, y = 0.0;
        for (int s = 0; s < rows; s++) {
            const long i = (long)m * rows + s;
            id[i] = (double)m;
            times[i] = (double)(i + 1);
            x += lcg(&seed) - 0.5; y += lcg(&seed) - 0.5;
            obs[i] = x + 0.1 * (lcg(&seed) - 0.5);
            obs[i + n] = y + 0.1 * (lcg(&seed) - 0.5);
            for (int k = 0; k < K; k++) xre[i + (long)k * n] = sin(0.05 * (double)(s + 1) * (double)(k + 1)) / (double)(k + 1);
        }
    }
    int32_t ncol_fe[4] = {1, 1, 1, 1}, ncol_re[4] = {0, 0, 0, 0}, smooth_ncol[1] = {0};
    const double *x_fe[4] = {NULL, NULL, NULL, NULL}, *x_re[4] = {NULL, NULL, NULL, NULL};
    double S[9] = {2, -1, 0, -1, 2, -1, 0, -1, 2};
    ssde_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.abi_version = SSDE_ABI_VERSION;
    desc.model = SSDE_MODEL_CTCRW;
    desc.n_dim = d; desc.n_par = q; desc.n = n;
    desc.id = id; desc.times = times; desc.obs = obs;
    desc.ncol_fe = ncol_fe; desc.x_fe = x_fe; desc.ncol_re = ncol_re; desc.x_re = x_re;
    if (K > 0) {
        ncol_re[2] = K; x_re[2] = xre;                 /* a smooth on log tau */
        smooth_ncol[0] = K;
        desc.n_smooth = 1; desc.smooth_ncol = smooth_ncol; desc.s_blocks = S;
    }
    desc.include_penalty = 1;
    desc.na_mode = SSDE_NA_ANY_NAN;
    desc.device = -1;
    ssde_handle *h = NULL;
    int st = ssde_create(&desc, &h);
    if (st != SSDE_OK) {
        printf("create failed: %d %s\n", st, ssde_last_error(NULL));
        return st == SSDE_ERR_NODEVICE ? 3 : 4;
    }
    ssde_info_t inf;
    ssde_info(h, &inf);
    const int np = inf.n_par_full;                     /* [log_sigma_obs | 4 intercepts | log_lambda | K coeff_re] */
    double *par = calloc((size_t)np, sizeof(double)), *grad = calloc((size_t)np, sizeof(double));
    par[0] = log(0.2); par[3] = 0.3; par[4] = -0.1;
    for (int k = 0; k < K; k++) par[np - K + k] = 0.05 * (double)(k + 1);
    double v0 = 0.0, v1 = 0.0;
    if (ssde_eval(h, par, np, 0, &v0, NULL) != SSDE_OK || ssde_eval(h, par, np, 1, &v1, grad) != SSDE_OK) {
        printf("eval failed: %s\n", ssde_last_error(h));
        return 5;
    }
    printf("%.17g", v1);
    for (int k = 0; k < np; k++) printf(" %.17g", grad[k]);
    double lv = 0.0;
    double *par2 = malloc(sizeof(double) * (size_t)np), *g2 = calloc((size_t)np, sizeof(double));
    memcpy(par2, par, sizeof(double) * (size_t)np);
    if (ssde_laplace_eval(h, par2, np, 1, &lv, g2, NULL, NULL) != SSDE_OK) { printf("\nlaplace failed: %s\n", ssde_last_error(h)); return 6; }
    double *aest = malloc(sizeof(double) * (size_t)n * (size_t)inf.sdim);
    if (ssde_report(h, par, np, aest) != SSDE_OK) { printf("\nreport failed: %s\n", ssde_last_error(h)); return 7; }
    ssde_info(h, &inf);
    printf(" | %.17g | %lld %lld | %.17g %d\n", lv, (long long)inf.n_evals, (long long)inf.n_memo_hits, aest[n - 1], v0 == v1);
    ssde_destroy(h);
    free(id); free(times); free(obs); free(xre); free(par); free(grad); free(par2); free(g2); free(aest);
    return 0;
}
