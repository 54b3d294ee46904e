/*
 * CPU oracle, plain C: a restatement of PyRad's hot loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; nothing
 * under pyrad_amd/ does.  Parity status: pinned — tests/test_oracle_golden.py checks this
 * code against the golden vectors captured from the real reference
 * (tests/golden/make_golden.py) and against oracle/pyrad_oracle.py.
 *
 * Restates Isotope.createCrossSection (pyradClasses.py:361-400) in the reference's loop
 * order: for each line, evaluate the right half-profile on arange(0, dfc, res)
 * (pyradLineshape.py:39, 52, 58-76), correct the intensity (pyradIntensity.py:16-32), then
 * add the centre and mirror both wings one grid point at a time with a bounds test per point
 * (pyradClasses.py:392-400).  The regrid of pyradClasses.py:401-405 stays in NumPy
 * (oracle/pyrad_oracle.py regrid_to_base).  Single-threaded, like the reference.
 *
 * Build:  gcc -O2 -fPIC -shared -fno-fast-math -ffp-contract=off oracle/lbl_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static const double k_B = 1.38064852E-23;      /* pyradClasses.py:16 */
static const double c_light = 299792458.0;     /* pyradClasses.py:15 */
static const double h_planck = 6.62607004e-34; /* pyradClasses.py:18 */
static const double pi_ = 3.141592653589793;   /* pyradClasses.py:19 */
static const double t0 = 296.0, p0 = 1013.25;  /* pyradClasses.py:20-21 */
static const double avo = 6.022140857E23;      /* pyradClasses.py:23 */

/* returns the exact number of (line, grid point) adds performed, or -1 on allocation failure */
int64_t lbl_oracle_xsec(const double* nu, const double* sw, const double* elower, const double* gamma_air,
                        const double* gamma_self, const double* n_air, const double* delta_air, int64_t n_lines,
                        double T, double P, double q, double molmass, double Q_T, double Q_296,
                        double range_min, double resolution, int64_t W, int64_t n_work,
                        double* cross_section /* n_work, zeroed by the caller */, int64_t regime_counts[3]) {
    const double c2 = c_light * h_planck * 100 / k_B;               /* pyradIntensity.py:13 */
    double* x = (double*)malloc((size_t)(W > 0 ? W : 1) * sizeof(double));
    double* curve = (double*)malloc((size_t)(W > 0 ? W : 1) * sizeof(double));
    if (!x || !curve) { free(x); free(curve); return -1; }
    int64_t evals = 0;
    for (int64_t i = 0; i < W; ++i) x[i] = 0 + (double)i * resolution;   /* np.arange(0, dfc, res), cls:377 */
    regime_counts[0] = regime_counts[1] = regime_counts[2] = 0;
    const double m = molmass / 1000 / avo;                            /* pyradClasses.py:296 */
    for (int64_t l = 0; l < n_lines; ++l) {
        const double broadened = nu[l] + delta_air[l] * P / p0;       /* pyradClasses.py:254 */
        const double lhw = ((1 - q) * gamma_air[l] + q * gamma_self[l]) * (P / p0) * pow(t0 / T, n_air[l]); /* :258-259 */
        const double ghw = broadened * sqrt(2 * k_B * T / m / (c_light * c_light));   /* pyradClasses.py:263 */
        const double ratio = lhw / ghw;                                /* pyradClasses.py:378 */
        if (ratio < .01) {                                             /* pyradClasses.py:379-381 */
            for (int64_t i = 0; i < W; ++i)
                curve[i] = exp(-(x[i] * x[i]) / (ghw * ghw)) / ghw / sqrt(pi_);          /* pyradLineshape.py:39 */
            regime_counts[0]++;
        } else if (ratio > 100) {                                      /* pyradClasses.py:382-384 */
            for (int64_t i = 0; i < W; ++i)
                curve[i] = lhw / pi_ / (x[i] * x[i] + lhw * lhw);                         /* pyradLineshape.py:52 */
            regime_counts[1]++;
        } else {                                                       /* pyradLineshape.py:58-76 */
            const double g = 2 * ghw, lf = 2 * lhw;
            const double f = pow(pow(g, 5) + 2.69269 * pow(g, 4) * lf + 2.42843 * pow(g, 3) * (lf * lf) +
                                 4.47163 * (g * g) * pow(lf, 3) + .07842 * g * pow(lf, 4) + pow(lf, 5), .2);
            const double r = lf / f;
            const double eta = 1.36603 * r - .47719 * (r * r) + .11116 * pow(r, 3);
            const double hw = f / 2;
            for (int64_t i = 0; i < W; ++i) {
                const double gc = exp(-(x[i] * x[i]) / (hw * hw)) / hw / sqrt(pi_);
                const double lc = hw / pi_ / (x[i] * x[i] + hw * hw);
                curve[i] = eta * lc + (1 - eta) * gc;
            }
            regime_counts[2]++;
        }
        /* pyradIntensity.intensityFactor (pyradIntensity.py:16-32) at the shifted wavenumber (cls:388) */
        const double stim = (1 - exp(-c2 * broadened / T)) / (1 - exp(-c2 * broadened / t0));
        const double boltz = exp(-c2 * elower[l] / T) / exp(-c2 * elower[l] / t0);
        const double intensity = sw[l] * (Q_296 / Q_T) * stim * boltz;
        const int64_t idx = (int64_t)((nu[l] - range_min) / resolution);   /* pyradClasses.py:390, trunc */
        const int64_t last = n_work - 1;                                   /* pyradClasses.py:391 */
        if (idx >= 0 && idx <= last) { cross_section[idx] = cross_section[idx] + curve[0] * intensity; evals++; }
        for (int64_t dx = 1; dx < W - 1; ++dx) {                           /* pyradClasses.py:394 */
            const int64_t ri = idx + dx, li = idx - dx;
            if (ri >= 0 && ri <= last) { cross_section[ri] += curve[dx] * intensity; evals++; }   /* :397-398 */
            if (li >= 0 && li <= last) { cross_section[li] += curve[dx] * intensity; evals++; }   /* :399-400 */
        }
    }
    free(x);
    free(curve);
    return evals;
}
