// Hand-written CDNA4 (gfx950) kernels of the line-by-line absorption engine.
//
//   K1 line_prep_kernel        pyradClasses.py:252-263, 378-390; pyradLineshape.py:59-71;
//                              pyradIntensity.py:16-32
//   K2 xsec_accumulate_*       pyradLineshape.py:39, 52, 72-74 and the scatter loop
//                              pyradClasses.py:392-400, restated as an owner-computes gather
//   K3 regrid_kernel           np.interp of pyradClasses.py:401-405 (only when res != BASE)
//   K4 layer_sweep_kernel      pyradClasses.py:566-571, 583, 707-716, 784-787; pyradPlanck.py:38-44
//   K5 column_sweep_kernel     fold of pyradClasses.py:784-787 over layers
//   K5b column_step_kernel     K4's arithmetic per layer + that fold, straight from the cross sections
//   K6 band_integral kernels   pyradClasses.py:26-29
//   K7 line_survey_kernel      pyradClasses.py:409-428
//
// Design notes (DESIGN.md has the long form).  The reference snaps every line centre to a
// grid index and samples the half-profile at integer multiples of the resolution, so the
// contribution of line l to grid point j depends only on |j - c_l|.  That makes the gather
// form exact: every lane owns R consecutive grid points in registers, a wavefront walks the
// (sorted) lines whose support reaches its 64*R points, and each grid point is written once
// with a plain coalesced store.  No atomics, and a fixed summation order per grid point, so two
// runs are bit-identical.
//
// K2 variants (lbl_set_option "accum_variant"; 1, 2 and 4 exist in diagnostic builds only, -DLBL_DIAG):
//   0    xsec_accumulate_kernel      the literal form: IEEE divide + exp per pair, records through the scalar cache (on-device cross-check)
//   1-2  xsec_accumulate_kernel      running fraction / + Gaussian recurrence through the scalar cache (superseded)
//   3    xsec_accumulate_lds_kernel  records streamed through wave-private LDS, every pair direct
//   4    ..._balanced_kernel         3 with an exactly balanced partition of (span, line) pairs (superseded)
//   5    xsec_accumulate_lds_kernel<.., FF = true>  (default) 3 + far-field series for distant
//        Lorentz lines; optionally the layer sweep of a single-line-list layer in the output stage;
//        round 3: its edge lines (support ends inside the span) take the skewed walk (skew_edges), and
//        line lists whose window is too narrow for any far line (< 640 points: the upper layers of a
//        column) run xsec_accumulate_skew_kernel, in which every LANE walks the lines that reach its
//        own points (lbl_set_option "accum_skew")
//
// Round 5: an accumulate job may hold ALL line lists of a layer (merged layer job: lbl_layer_merged_step_dev,
// lbl_layers_merged_accumulate_dev).  K1 (line_prep_merged_kernel) then writes the lists' records into one array in
// centre-index order with every amplitude times its molecule's conc P / 1E4 / k / T over a power of two, and the same K2
// kernels accumulate the layer's absorption coefficient sum_m f_m sum_iso xs_iso (pyradClasses.py:707-712, 581-583, 566-571)
// directly - the "shared wavenumber-grid absorption-coefficient array" - with the sweep in the output stage (output_point).
//
// K2 is fp64-VALU bound, not HBM bound: its compulsory traffic is 56 B per line and 8 B per grid
// point against 5 fp64 instructions per directly evaluated (line, grid point) pair.
#include "lbl_device.h"
#include "lbl_launch_shapes.h"
#include <cstdlib>
#include <type_traits>

namespace lbl {

// Timing-only ablations (parts of a kernel switched off: WRONG results) exist in diagnostic builds only
// (make EXTRA=-DLBL_DIAG, scripts/make_diag_lib.sh); the production library carries none of them.
#ifdef LBL_DIAG
#define LBL_ABLATE(obj, bit) (((obj).ablate & (bit)) != 0)
#else
#define LBL_ABLATE(obj, bit) false
#endif

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Prepared line records are written by K1 and only read by K2.  The scalar-cache variants
// view them through the constant address space (invariant memory -> s_load with a uniform
// index); the LDS variant streams them with coalesced vector loads.
typedef const double __attribute__((address_space(4)))* ConstF64;
typedef const int32_t __attribute__((address_space(4)))* ConstI32;

// working copy of one line (hot + cold halves) in registers
struct Rec {
    double cf, a2, KL, KG, b, q2;
    int32_t ci, dgi, flags;
};

__device__ __forceinline__ Rec load_rec_scalar(const HotRec* hot, const ColdRec* cold, int i, bool want_cold) {
    ConstF64 h = (ConstF64)(unsigned long long)hot + (long long)i * 4;
    ConstI32 hi = (ConstI32)(h + 3);
    Rec r;
    r.cf = h[0]; r.a2 = h[1]; r.KL = h[2]; r.dgi = hi[0]; r.flags = hi[1];
    r.ci = (int32_t)r.cf;
    r.KG = 0.0; r.b = 0.0; r.q2 = -1.0;
    if (want_cold) {
        ConstF64 c = (ConstF64)(unsigned long long)cold + (long long)i * 4;
        r.KG = c[0]; r.b = c[1]; r.q2 = c[2];
    }
    return r;
}

// xAxis element j of np.linspace(start, stop, n, endpoint=True) (pyradClasses.py:702-705):
// NumPy computes arange(n) * step + start with a separate multiply and add and then
// overwrites the last element with `stop`, so contraction to an FMA is switched off here.
__device__ __forceinline__ double linspace_at(long long j, long long n, double start, double stop, double step) {
#pragma clang fp contract(off)
    if (n > 1 && j == n - 1) return stop;
    double y = (double)j * step;
    return y + start;
}

// pyradPlanck.planckWavenumber (pyradPlanck.py:38-44): a / (exp(b) - 1),
// a = 2E8*h*c**2 * n**3, b = 100*h*c*n/k/T.  pa = 2E8*h*c**2 and pb = 100*h*c come from the
// host in the reference's association order.
// x / c for a launch-uniform divisor, bit for bit the IEEE quotient the reference's NumPy expression
// produces, in 5 instructions instead of the ~14 of the general divide (the sweeps are ALU bound on
// these: 15 fp64 divisions per grid point).  rc = RN(1/c) from the host.  q0 = RN(x rc) is within
// (1 + 2^-52) ulp of x/c; r = x - q c is formed by one fma; q1 = RN(q0 + r rc) is a faithful
// quotient; by Markstein's theorem one more correction with an exact residual gives RN(x/c) for every
// x whenever rc is the correctly rounded reciprocal of a c whose significand is not all ones.  The host
// passes rc = 0 for such a c (or a subnormal / non-finite one) and the plain divide is taken instead.
// Range: the proof needs the residual x - q c to be exact, i.e. not to fall below the subnormal grid:
// |x| >= 2^-969 (about 1e-292) for the divisors used here; below that the quotient can be off by one ulp
// of a number that is itself below 1e-270, and x = +-inf gives NaN (inf - inf) where IEEE gives +-inf.
// Cross sections times volume fractions times pressures are many orders of magnitude inside the range
// (the smallest non-zero cross section a line list produces here is ~1e-200: Gaussian tails underflow to
// exact zeros first, and 0 is handled exactly); a guard per call would cost the sweeps 6 % of their
// instructions.  tests/test_gpu_abi.py::test_sweep_divisions_are_ieee_exact covers 1e-270 .. 1e+250 and 0.
__device__ __forceinline__ double div_uniform(double x, double c, double rc) {
    if (rc == 0.0) return x / c;
    double q = x * rc;
    double r = fma(-q, c, x);
    q = fma(r, rc, q);
    r = fma(-q, c, x);
    return fma(r, rc, q);
}
constexpr double kRcp1E4 = 1.0 / 1E4;      // correctly rounded by the compiler; neither significand is all ones
constexpr double kRcpKB = 1.0 / kB;

// exp(x) for x <= 700 without the range tests of the library routine (22 -> 17 instructions); exact
// to the library's accuracy for x >= -745, 0 (through v_ldexp_f64) below, finite garbage-free
// down to about -1e18, so callers clamp only when their argument can be more extreme:
// n = rint(x log2 e), r = x - n ln2 (two-part), degree-11 polynomial on |r| <= ln2/2 (the
// coefficients of the ROCm device library's double-precision exp), scaled by 2^n with v_ldexp_f64,
// which also delivers the gradual underflow.  Callers clamp the argument.
__device__ __forceinline__ double exp_clamped(double x) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fefa39efp-1, x);
    r = fma(n, -0x1.abc9e3b39803fp-56, r);
    double p = fma(0x1.ade156a5dcb37p-26, r, 0x1.28af3fca7ab0cp-22);
    p = fma(p, r, 0x1.71dee623fde64p-19);
    p = fma(p, r, 0x1.a01997c89e6b0p-16);
    p = fma(p, r, 0x1.a01a014761f6ep-13);
    p = fma(p, r, 0x1.6c16c1852b7b0p-10);
    p = fma(p, r, 0x1.1111111122322p-7);
    p = fma(p, r, 0x1.55555555502a1p-5);
    p = fma(p, r, 0x1.5555555555511p-3);
    p = fma(p, r, 0x1.000000000000bp-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// the part that depends on the grid point only (a, and pb * n / kB): hoisted out of loops over temperatures
__device__ __forceinline__ void planck_point(double n, double pa, double pb, double& a, double& bn) {
    a = pa * (n * n * n);
    bn = div_uniform(pb * n, kB, kRcpKB);
}
__device__ __forceinline__ double planck_at(double a, double bn, double T, double rT) {
    const double b = div_uniform(bn, T, rT);                              // pb * n / kB / T
    return a / (exp(b) - 1.0);
}
__device__ __forceinline__ double planck_wn(double n, double T, double rT, double pa, double pb) {
    double a, bn;
    planck_point(n, pa, pb, a, bn);
    return planck_at(a, bn, T, rT);
}

// xs * conc * P / 1E4 / kB / T (pyradClasses.py:583), left to right like the reference
__device__ __forceinline__ double abs_coef_term(double xs, double conc, double P, double T, double rT) {
#pragma clang fp contract(off)
    return div_uniform(div_uniform(div_uniform(xs * conc * P, 1E4, kRcp1E4), kB, kRcpKB), T, rT);
}

// ---- the sweeps' default arithmetic (lbl_set_option "sweep_ieee_divisions" 0; flag `budget` in the argument blocks) ----
// The reference's NumPy expressions round nine times per point and layer where it does not matter: crossSection *
// concentration * P / 1E4 / k / T is three correctly rounded divisions (51 of the column step's 161 instructions with
// div_uniform), and the Planck function two more plus the library exp.  The cross section entering them already differs
// from NumPy's in its last bits (K2 is 1e-14, not bit-exact), so the chain buys nothing a comparison with the reference can
// see.  By default the molecule's factor conc * P / 1E4 / k / T comes from the host (evaluated there in the reference's
// order) and meets the cross section in ONE multiplication; the Planck exponent is n * (100 h c / k / T) with the bracket
// from the host; reciprocals by v_rcp_f64 + two Newton steps; exp without the library's range tests.  Each result is within
// a few 1e-16 of the chain's (tests/test_gpu_abi.py::test_sweep_divisions_are_ieee_exact checks both).  "sweep_ieee_divisions"
// 1 selects the chain: k is then bit-identical to NumPy's expression applied to the same cross section.
__device__ __forceinline__ double rcp_newton(double x) {           // x finite, positive, normal
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// (a NaN exponent must stay a NaN, as in np.exp(nan): fmin / fmax drop it, so it is put back explicitly - advisor, round 4)
__device__ __forceinline__ double planck_budget(double n, double pa, double pbkT) {
    const double b = n * pbkT;
    const double e = exp_clamped(fmin(b, 700.0)) - 1.0;
    const double v = (pa * (n * n * n)) * rcp_newton(fmax(e, 1e-300));
    const double r = (b > 700.0 || !(e > 0.0)) ? ((b > 700.0) ? 0.0 : (pa * (n * n * n)) / e) : v;     // (n = 0: 0/0 like the reference)
    return b != b ? b : r;
}
__device__ __forceinline__ double exp_neg_budget(double x) {       // exp(-x), x >= 0 (optical depth)
    // (the clamp as a select, not as fmax, which would drop a NaN: a NaN argument then runs through exp_clamped as a NaN -
    // rint, fma, ldexp all pass it on - and needs no test of its own; x = +inf gives 0 like the reference's exp(-inf))
    return exp_clamped(x > 800.0 ? -800.0 : -x);
}

// ----------------------------------------------------------------------------------------
// K1: per-line preparation
// ----------------------------------------------------------------------------------------
// What the reference computes per line before it touches the grid (pyradClasses.py:252-263, 378, 388-390;
// pyradIntensity.py:16-32): shared by K1 and by the kernel behind lbl_line_quantities, which only reports them.
struct LinePhysics {
    double broadened, lhw, ghw, ratio, A, fidx;
};
__device__ __forceinline__ LinePhysics line_physics(const PrepJob& J, int i) {
    LinePhysics L;
    const double nu = J.nu[i];
    const double q = J.q_frac;
    const double Pp0 = J.P_over_p0;                                   // P / p0, evaluated on the host
    // Line.broadenedLine (pyradClasses.py:252-254)
    L.broadened = nu + J.delta_air[i] * J.P / p0;
    // Line.lorentzHW (pyradClasses.py:256-259)
    // (t0/T)**n evaluated as exp(n ln(t0/T)) with the logarithm hoisted to the host (one per job)
    L.lhw = ((1.0 - q) * J.gamma_air[i] + q * J.gamma_self[i]) * Pp0 * exp(J.n_air[i] * J.log_t0_over_T);
    // Isotope.molMass (pyradClasses.py:294-296), Line.gaussianHW (pyradClasses.py:261-263): the
    // square root depends on the job only (host)
    L.ghw = L.broadened * J.ghw_factor;
    L.ratio = L.lhw / L.ghw;                                          // pyradClasses.py:378
    // pyradIntensity.intensityFactor (pyradIntensity.py:16-32) at the SHIFTED wavenumber
    // (pyradClasses.py:388).  Divisions by the per-job T and t0 are multiplications by their
    // reciprocals: at most one ulp in an exponent of order 1-50.
    const double c2 = cLight * hPlanck * 100.0 / kB;                  // pyradIntensity.py:13
    const double E = J.elower[i];
    const double stim = (1.0 - exp(-c2 * L.broadened * J.inv_T)) / (1.0 - exp(-c2 * L.broadened * (1.0 / t0)));
    // exp(-c2 E/T) / exp(-c2 E/t0) as one exponential of the difference (exactly 1 at T = t0,
    // like the quotient; elsewhere within |c2 E (1/T - 1/t0)| ulps of it, < 1e-14 relative)
    const double boltz = exp(c2 * E * (1.0 / t0) - c2 * E * J.inv_T);
    L.A = J.sw[i] * J.q_ratio * stim * boltz;
    // centre index from the UNSHIFTED wavenumber, truncation toward zero (pyradClasses.py:390)
    L.fidx = (nu - J.range_min) / J.resolution;
    return L;
}
__device__ __forceinline__ int line_regime(double ratio) {          // pyradClasses.py:379-387
    return ratio < .01 ? 0 : (ratio > 100.0 ? 1 : 2);
}

// One line's records (hot, cold), centre index and regime from its list's job constants: shared by the per-list K1 and
// the merged-order K1 below.
__device__ __forceinline__ void prep_one_line(const PrepJob& J, int i, HotRec& r, ColdRec& rc, long long& idx, int& regime) {
    const LinePhysics L = line_physics(J, i);
    // (merged layer job: the molecule's conc P / 1E4 / k / T over a power of two rides on the intensity; 1.0 otherwise: exact)
    const double lhw = L.lhw, ghw = L.ghw, ratio = L.ratio, A = L.A * J.weight;
    idx = (long long)L.fidx;
    if (idx > 2000000000LL) idx = 2000000000LL;
    if (idx < -2000000000LL) idx = -2000000000LL;

    double hw, KL, KG;
    if (ratio < .01) {                // Gaussian only (pyradClasses.py:379-381)
        regime = 0;
        hw = ghw;
        KL = 0.0;
        KG = A / hw * kInvSqrtPi;                                     // pyradLineshape.py:39 (/ sqrt(pi) as a product)
    } else if (ratio > 100.0) {       // Lorentz only (pyradClasses.py:382-384)
        regime = 1;
        hw = lhw;
        KL = A * (hw * kInvPi);                                       // pyradLineshape.py:52 (/ pi as a product)
        KG = 0.0;
    } else {                          // pseudo-Voigt (pyradClasses.py:385-387, pyradLineshape.py:58-76)
        regime = 2;
        const double g = 2.0 * ghw, l = 2.0 * lhw;
        const double g2 = g * g, l2 = l * l;
        const double f5 = g2 * g2 * g + 2.69269 * g2 * g2 * l + 2.42843 * g2 * g * l2 +
                          4.47163 * g2 * l2 * l + .07842 * g * l2 * l2 + l2 * l2 * l;
        const double f = exp(.2 * log(f5));                                 // f5 ** .2
        const double x = l / f;
        const double eta = 1.36603 * x - .47719 * x * x + .11116 * x * x * x;
        hw = f / 2.0;
        KL = eta * (A * (hw * kInvPi));
        KG = (1.0 - eta) * (A / hw * kInvSqrtPi);
    }
    const double a = hw * J.inv_res;
    r.cf = (double)idx;
    r.a2 = a * a;
    r.KL = KL * J.inv_res2;
    rc.KG = KG;
    rc.b = 1.0 / r.a2;
    r.flags = 0;
    // running-fraction accumulation multiplies up to 32 denominators d*d + a2 with
    // |d| <= H + 512 <= 4e4 (else 16, see AccumJob.flush_every): keep a2 in [1e-9, 1e8] so
    // the product stays within 1e-288 .. 1e296; anything else takes the plain-divide path
    if (!(r.a2 > 1e-9 && r.a2 < 1e8)) r.flags |= REC_DIRECT_DIV;
    // Gaussian term: where can it still change the fp64 value of the line's sum?
    double dg = 0.0;
    if (KG != 0.0) {
        const double u2_under = 745.2;           // exp(-745.2) == 0 in fp64
        double u2 = u2_under;
        if (KL != 0.0) {
            // ratio Gauss/Lorentz at offset u = d/a:  C (1+u^2) exp(-u^2);  solve = 2^-54 (budget mode: 2^-34) for u^2 = v + 1,
            // v = ln C + ln(1 + v).  The cut-off only has to err on the far side, so single precision
            // with a margin does: ln C from the exponent and a hardware log2 of the mantissa, three
            // fixed-point steps (each contracts by 1/(1+v)), +0.01 for the float roundings and the
            // remaining contraction (4 double-precision logs were a fifth of this kernel's instructions).
            const double C = fabs(KG / (r.KL * rc.b)) * J.gauss_cut;        // 2^54 (exact mode) or 2^34 (budget mode)
            if (C <= 1.0) {
                u2 = 0.0;
            } else {
                int ex;
                const float mant = (float)frexp(C, &ex);                               // C = mant 2^ex, mant in [0.5, 1)
                const float lnC = ((float)ex + __log2f(mant)) * 0.69314718f;
                float v = lnC;
                for (int it = 0; it < 3; ++it) v = lnC + __log2f(1.0f + v) * 0.69314718f;
                u2 = fmin((double)(v * 1.00001f + 1.01f), u2_under);
            }
        }
        dg = (u2 > 0.0) ? (double)(__fsqrt_rn((float)u2) * 1.000001f) * a + 2.0 : 0.0;
    }
    r.dgi = (dg < 2.0e9) ? (int32_t)dg : 2000000000;
#ifdef LBL_DIAG
    if (J.pad & 1024) r.dgi = 0;               // (timing only, debug_ablate 1024: no Gaussian part anywhere)
#endif
    // Gaussian recurrence along a lane's consecutive points: only for b <= 4 (see gauss_term);
    // narrower profiles take one exp per point
    rc.q2 = (rc.b <= 4.0) ? exp(-2.0 * rc.b) : -1.0;
    if (!(rc.b <= 4.0)) r.flags |= REC_NO_RECUR;
    // 16-point runs (gauss_runs): a lane walks 15 steps from its first point, possibly TOWARDS the centre.
    // If its seed KG exp(-b d0^2) has underflowed (b d0^2 > T, T = 745 - ln(1/KG) >= ~600 for any KG down
    // to 1e-60) the run's values stay 0; that is harmless as long as no point of the run can matter: the
    // nearest one has b d^2 > b (sqrt(T/b) - 15)^2, which exceeds the 45 beyond which the term is below
    // 2^-54 of the line's Lorentz part whenever sqrt(b) < (sqrt(600) - sqrt(45)) / 15 = 1.19.  b <= 1
    // (profiles at least one grid point wide) keeps a margin.  (Seeds that are tiny but normal keep the
    // recurrence exact; a clamped r only lowers a value that is negligible anyway.)  Pure-Gaussian lines
    // (no Lorentz part to be negligible against: their term counts until it underflows) keep the 4-point pass.
    if (rc.b <= 1.0 && KL != 0.0) r.flags |= REC_LONG_RUN;
    // 32-point runs (round 6, the three-waves-per-SIMD build of the far-field kernel): 31 steps towards the centre need
    // sqrt(b) < (sqrt(600) - sqrt(45)) / 31 = 0.57; b <= 0.3 keeps the same margin
    if (rc.b <= 0.3 && KL != 0.0) r.flags |= REC_LONG_RUN32;
    rc.KLd = r.KL;
}

__global__ __launch_bounds__(256) void line_prep_kernel(const PrepJob* __restrict__ jobs) {
    const PrepJob& J = jobs[blockIdx.y];
    if (J.merged) return;                       // (its job's merged-order launch prepares this list)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int regime = -1;
    if (i < J.n_lines) {
        HotRec r;
        ColdRec rc;
        long long idx;
        prep_one_line(J, i, r, rc, idx, regime);
        J.hot[i] = r;
        J.cold[i] = rc;
        J.cidx[i] = (int32_t)idx;
    }
    // regime counters (pyradClasses.py:368-370, 406).  One plain store per block: thousands of
    // atomics on one cache line cost ~12 ns each and made this kernel 3x longer than its arithmetic.
    __shared__ unsigned int s_cnt[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < 3; ++k) {
        const unsigned long long m = __ballot(regime == k);
        if (lane == 0) s_cnt[wave][k] = (unsigned int)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 3)
        J.block_counts[blockIdx.x * 3 + threadIdx.x] =
            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];
}

// K1 of a merged layer job, in MERGED order: thread t prepares the line that belongs at position t of the layer's record
// array (src[t] = list within the job << 26 | line within the list, built once per window by merge_rank_kernel), so the
// records are written with coalesced stores; the seven HITRAN fields are gathered from the job's lists, each of which
// is walked in increasing order.  (The first version had every list scatter its records through the inverse map: the
// interleaved 32-byte stores of 90 lists made K1 of the 30-layer column twice as long, 0.39 against 0.20 ms.)
__global__ __launch_bounds__(256) void line_prep_merged_kernel(const PrepJob* __restrict__ lists, const MergedPrep* __restrict__ jobs) {
    const MergedPrep& M = jobs[blockIdx.y];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.x >= M.blocks) return;
    // the job's per-list constants and field pointers, staged in LDS: every lane picks its own list's block (neighbouring
    // merged positions belong to different lists), which as global loads were ~25 divergent fetches per line
    __shared__ PrepJob s_lists[kMaxIso];            // (a merged job holds at most kMaxIso lists: enqueue_accumulate)
    {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(lists + M.first_list);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(s_lists);
        const int words = M.n_lists * (int)(sizeof(PrepJob) / 8);
        for (int w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
        __syncthreads();
    }
    int regime = -1, list = -1;
    if (t < M.n_total) {
        const int s = M.src[t];
        list = (int)((unsigned int)s >> 26);
        const PrepJob& J = s_lists[list];
        HotRec r;
        ColdRec rc;
        long long idx;
        prep_one_line(J, s & ((1 << 26) - 1), r, rc, idx, regime);
        M.hot[t] = r;
        M.cold[t] = rc;
        M.cidx[t] = (int32_t)idx;
    }
    // regime counters per list and block of 256 merged positions
    __shared__ unsigned int s_cnt[4][kMaxIso][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int l = 0; l < M.n_lists; ++l)
        for (int k = 0; k < 3; ++k) {
            const unsigned long long m = __ballot(list == l && regime == k);
            if (lane == 0) s_cnt[wave][l][k] = (unsigned int)__popcll(m);
        }
    __syncthreads();
    for (int q = threadIdx.x; q < M.n_lists * 3; q += blockDim.x) {
        const int l = q / 3, k = q % 3;
        lists[M.first_list + l].block_counts[blockIdx.x * 3 + k] = s_cnt[0][l][k] + s_cnt[1][l][k] + s_cnt[2][l][k] + s_cnt[3][l][k];
    }
}

// lbl_line_quantities: the per-line numbers of the reference (Line.lorentzHW / gaussianHW, the corrected intensity,
// the regime) from the same expressions K1 evaluates, and the centre index K1 itself wrote (the one K2 works
// from; K1 clamps it to +-2e9).  Never part of a step.
__global__ __launch_bounds__(256) void line_quantities_kernel(const PrepJob* __restrict__ jobs, long long* __restrict__ index,
                                                              double* __restrict__ lhw, double* __restrict__ ghw,
                                                              double* __restrict__ intensity, int32_t* __restrict__ regime) {
    const PrepJob& J = jobs[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= J.n_lines) return;
    const LinePhysics L = line_physics(J, i);
    index[i] = (long long)J.cidx[i];
    lhw[i] = L.lhw; ghw[i] = L.ghw; intensity[i] = L.A;
    regime[i] = line_regime(L.ratio);
}

// Merged layer jobs: which line of which list belongs at every position of the layer's one record array (centre-index
// order, ties by list).  Built once per (line lists, grid) beside the dispatch schedule and kept with it; K1 then runs in
// merged order (line_prep_merged_kernel) in every step.  centre_index_kernel evaluates K1's own expression (line_physics / line_prep_kernel:
// (nu - range_min) / resolution, truncated, clamped), so the order is the order of the very indices K1 will write.
__global__ __launch_bounds__(256) void centre_index_kernel(const MergeList* __restrict__ lists) {
    const MergeList& M = lists[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_lines) return;
    long long idx = (long long)((M.nu[i] - M.range_min) / M.resolution);
    if (idx > 2000000000LL) idx = 2000000000LL;
    if (idx < -2000000000LL) idx = -2000000000LL;
    M.tmp_cidx[i] = (int32_t)idx;
}

__global__ __launch_bounds__(256) void merge_rank_kernel(const MergeList* __restrict__ lists) {
    const MergeList& M = lists[blockIdx.y];
    const int i0 = blockIdx.x * blockDim.x;
    if (i0 >= M.n_lines) return;                                  // (whole workgroup)
    const int i = i0 + threadIdx.x;
    const int me = (int)blockIdx.y;
    // lists ahead of this one win ties (count their lines with c_b <= c), lists behind lose them (c_b < c).
    // Round 6: the workgroup's 256 lines are consecutive in a sorted list, so their places in another list lie between the
    // places of its first and of its last line.  Those two are found by two THREADS per other list, side by side (one search
    // deep, not two); the window between them - a few hundred entries - is staged in LDS with coalesced loads and searched
    // there.  Before: 34 scattered loads per line and other list, 400 M for a column, at the rate the L2 serves them (0.26 ms).
    constexpr int WIN = 2048;
    __shared__ int s_lo[kMaxIso], s_hi[kMaxIso];
    __shared__ int32_t s_win[WIN];
    auto search = [&](const int32_t* __restrict__ a, long long target, int lo, int hi) {
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)a[mid] < target) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    {
        const int u = (int)threadIdx.x & 63, which = (int)threadIdx.x >> 6;          // which 0: the first line's place, 1: the last line's
        const int b = M.job_first + u;
        if (which < 2 && u < M.job_count && b != me) {
            const MergeList& O = lists[b];
            const int i1 = min(i0 + (int)blockDim.x, M.n_lines) - 1;
            const long long tie = b < me ? 1 : 0;
            const int r = search(O.tmp_cidx, (long long)M.tmp_cidx[which ? i1 : i0] + tie, 0, O.n_lines);
            if (which) s_hi[u] = r; else s_lo[u] = r;
        }
    }
    __syncthreads();
    const bool mine = i < M.n_lines;
    const long long c = mine ? (long long)M.tmp_cidx[i] : 0;
    int pos = i;
    for (int b = M.job_first; b < M.job_first + M.job_count; ++b) {
        if (b == me) continue;
        const MergeList& O = lists[b];
        const int lo = s_lo[b - M.job_first], hi = s_hi[b - M.job_first];          // (uniform over the workgroup)
        const long long target = b < me ? c + 1 : c;
        if (hi - lo <= WIN) {
            __syncthreads();
            for (int k = threadIdx.x; k < hi - lo; k += blockDim.x) s_win[k] = O.tmp_cidx[lo + k];
            __syncthreads();
            if (mine) pos += lo + search(s_win, target, 0, hi - lo);
        } else if (mine) {
            pos += search(O.tmp_cidx, target, lo, hi);
        }
    }
    if (mine) M.src_of_job[pos] = (int32_t)(((unsigned int)(me - M.job_first) << 26) | (unsigned int)i);
}

void launch_merge_ranks(const MergeList* d_lists, int n_lists, int max_lines, hipStream_t s) {
    if (n_lists <= 0 || max_lines <= 0) return;
    dim3 grid((max_lines + 255) / 256, n_lists);
    hipLaunchKernelGGL(centre_index_kernel, grid, dim3(256), 0, s, d_lists);
    hipLaunchKernelGGL(merge_rank_kernel, grid, dim3(256), 0, s, d_lists);
}

// ----------------------------------------------------------------------------------------
// K2: owner-computes accumulation
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_i32(const int32_t* __restrict__ a, int n, long long target) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)a[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Gaussian part of one line for a lane's R consecutive points (d0 = offset of the lane's
// first point from the line centre), masked to the line's support |d| <= H.
//   GM == 0: one exp per point.
//   GM == 1: two exps per lane, then g(d+1) = g(d) r(d), r(d+1) = r(d) q2 with
//            r(d) = exp(-b (2d+1)), q2 = exp(-2b): 2 multiplies per further point.
// The recurrence keeps full relative precision while g(d0) is a normal number.  Walking
// TOWARDS the centre the terms grow by at most exp(b (R-1) (2|d0| - R + 1)); K1 enables the
// recurrence only for b <= 4 (q2 >= 0), for which a lane whose first point has underflowed
// (b d0^2 > 708) cannot reach a point where the term still matters (b d^2 < ~45) within R <= 8
// steps: there  b (d0^2 - d^2) <= 14 sqrt(45 b) + 49 b < 400.  The exponent of r is clamped so
// that a far lane computes 0 * finite = 0, never 0 * inf.
template <int R, int GM, bool MASKED = true>
__device__ __forceinline__ void gauss_term(double KG, double b, bool recur, double d0, double Hf, double q2,
                                           double (&acc)[R]) {
    if (GM == 0 || !recur || R < 4) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double t = KG * exp_clamped(fmax(-b * (d * d), -800.0));
            acc[k] += (!MASKED || fabs(d) <= Hf) ? t : 0.0;
        }
    } else {
        // b <= 4 and |d0| < 2^31 here: the arguments stay far inside the range where the reduction of
        // exp_clamped is finite (a hopeless argument only has to underflow to 0); only the growth of r
        // towards the centre needs its cap
        double g = KG * exp_clamped(-b * (d0 * d0));
        double rr = exp_clamped(fmin(-b * (2.0 * d0 + 1.0), 700.0));
#pragma unroll
        for (int k = 0; k < R; ++k) {
            if (MASKED) {
                const double d = d0 + (double)k;
                acc[k] += (fabs(d) <= Hf) ? g : 0.0;
            } else {
                acc[k] += g;
            }
            g *= rr;
            rr *= q2;
        }
    }
}

// Per-wave accumulator state.  acc: finished sums.  N/D: running fraction of the Lorentz
// terms of the last `cnt` (< 16) lines:  sum_i K_i/den_i = N/D with
//     N <- N*den + K*D,   D <- D*den        (3 fp64 ops + 2 for den instead of a divide)
// flushed into acc with one IEEE divide per point every `every` lines (32; 16 when the
// window is wider than 4e4 points).  K1 routes lines with a2 outside [1e-9, 1e8] to the
// plain-divide path, so the product of denominators stays within 1e-288 .. 1e296.
template <int R>
struct WaveAcc {
    double acc[R], N[R], D[R];
    int cnt, every;
    __device__ __forceinline__ void init(int flush_every) {
#pragma unroll
        for (int k = 0; k < R; ++k) { acc[k] = 0.0; N[k] = 0.0; D[k] = 1.0; }
        cnt = 0;
        every = flush_every;
    }
    // N/D by reciprocal + two Newton steps + one residual correction of the quotient (7 fp64
    // instructions instead of the 12 of an IEEE divide; D is a product of finite positive
    // denominators in [1e-288, 1e296], so no special cases; the quotient is within 1 ulp)
    __device__ __forceinline__ void flush() {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double r = __builtin_amdgcn_rcp(D[k]);
            r = fma(fma(-D[k], r, 1.0), r, r);
            r = fma(fma(-D[k], r, 1.0), r, r);
            double q = N[k] * r;
            q = fma(fma(-D[k], q, N[k]), r, q);
            acc[k] += q;
            N[k] = 0.0; D[k] = 1.0;
        }
        cnt = 0;
    }
};

// One line against a lane's R points.  DIV == 0: IEEE divide per pair; DIV == 2: running fraction.
template <int R, bool MASKED, int DIV>
__device__ __forceinline__ void lorentz_term(double a2, double KL, int flags, double d0, double Hf, WaveAcc<R>& S) {
    if (DIV == 0 || (flags & REC_DIRECT_DIV)) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            double t = KL / fma(d, d, a2);
            if (MASKED) t = (fabs(d) <= Hf) ? t : 0.0;
            S.acc[k] += t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double den = fma(d, d, a2);
            double K = KL;
            if (MASKED) K = (fabs(d) <= Hf) ? K : 0.0;
            const double t = K * S.D[k];
            S.N[k] = fma(S.N[k], den, t);
            S.D[k] *= den;
        }
        if (++S.cnt == S.every) S.flush();
    }
}

// ---- variants 0-2: line records through the scalar cache (one s_load per wave and line) ----
template <int R, bool MASKED, int DIV, int GM>
__device__ __forceinline__ void process_lines_scalar(const HotRec* hot, const ColdRec* cold, int i0, int i1, double x0,
                                                     int wlo, int whi, double Hf, WaveAcc<R>& S) {
    for (int i = i0; i < i1; ++i) {
        Rec r = load_rec_scalar(hot, cold, i, false);
        const double d0 = x0 - r.cf;
        lorentz_term<R, MASKED, DIV>(r.a2, r.KL, r.flags, d0, Hf, S);
        const int dist = max(0, max(r.ci - whi, wlo - r.ci));
        if (dist < r.dgi) {
            r = load_rec_scalar(hot, cold, i, true);
            gauss_term<R, GM>(r.KG, r.b, r.q2 >= 0.0, d0, Hf, r.q2, S.acc);
        }
    }
}

// Line ranges of a wave whose points span [wlo, whi] (cidx is sorted):
//   [iA, iB)  left-edge lines,   c in [wlo-H, whi-H)      -> masked
//   [iB, iC)  interior lines,    c in [whi-H, wlo+H]      -> every point inside the support
//   [iC, iD)  right-edge lines,  c in (wlo+H, whi+H]      -> masked
// Four lower bounds at once: lane group q = lane/16 searches target q with a 16-ary search
// (16 probes per step, range / 17 per step): 4-5 dependent loads for 10^5..10^6 lines instead
// of the 17-19 of a binary search, which cost a third of a short wave's lifetime.
__device__ __forceinline__ void wave_line_ranges(const int32_t* __restrict__ cidx, int n_lines, int wlo, int whi, int H,
                                                 int lane, int& iA, int& iB, int& iC, int& iD) {
    const long long tA = (long long)wlo - H, tB = (long long)whi - H;
    const long long tC = (long long)wlo + H + 1, tD = (long long)whi + H + 1;
    const int q = lane >> 4, jj = lane & 15;
    const long long tgt = q == 0 ? tA : q == 1 ? tB : q == 2 ? tC : tD;
    const unsigned long long gmask = 0xFFFFull << (q * 16);
    int lo = 0, hi = n_lines;                       // answer a = first i with cidx[i] >= tgt, a in [lo, hi]
    while (__any(hi > lo)) {
        const long long len = (long long)hi - lo;
        const int pos = lo + (int)(((long long)(jj + 1) * len) / 17);       // < hi whenever len > 0
        const bool below = (len > 0) && ((long long)cidx[len > 0 ? pos : 0] < tgt);
        const int k = __popcll(__ballot(below) & gmask);                     // probes are sorted: the first k are below
        if (len > 0) {
            const int p_k = lo + (int)(((long long)(k + 1) * len) / 17);
            const int p_km1 = lo + (int)(((long long)k * len) / 17);
            const int new_lo = k > 0 ? p_km1 + 1 : lo;
            const int new_hi = k < 16 ? p_k : hi;
            lo = new_lo; hi = new_hi;
        }
    }
    iA = __builtin_amdgcn_readlane(lo, 0);
    iB = __builtin_amdgcn_readlane(lo, 16);
    iC = __builtin_amdgcn_readlane(lo, 32);
    iD = __builtin_amdgcn_readlane(lo, 48);
    if (tB >= tC) { iB = iD; iC = iD; }      // span wider than the support: no interior line
}

// The far-field variant also needs the two indices that bound the NEAR lines, so six lower
// bounds: lane group q = lane/8 searches target q with a 9-way split per step (8 probes).
//   [iB, iF1)  far-left interior lines,  c <= fl      [iF2, iC)  far-right interior lines,  c >= fr
__device__ __forceinline__ void wave_line_ranges_far(const int32_t* __restrict__ cidx, int n_lines, int wlo, int whi, int H,
                                                     long long fl, long long fr, int lane, int& iA, int& iB, int& iC,
                                                     int& iD, int& iF1, int& iF2, int& iN1, int& iN2, long long nl, long long nr) {
    const long long tA = (long long)wlo - H, tB = (long long)whi - H;
    const long long tC = (long long)wlo + H + 1, tD = (long long)whi + H + 1;
    const int q = lane >> 3, jj = lane & 7;
    const long long tgt = q == 0 ? tA : q == 1 ? tB : q == 2 ? tC : q == 3 ? tD : q == 4 ? fl + 1 : q == 5 ? fr : q == 6 ? nl + 1 : nr;
    const unsigned long long gmask = 0xFFull << (q * 8);
    int lo = 0, hi = n_lines;
    while (__any(hi > lo)) {
        const long long len = (long long)hi - lo;
        const int pos = lo + (int)(((long long)(jj + 1) * len) / 9);
        const bool below = (len > 0) && ((long long)cidx[len > 0 ? pos : 0] < tgt);
        const int k = __popcll(__ballot(below) & gmask);
        if (len > 0) {
            const int p_k = lo + (int)(((long long)(k + 1) * len) / 9);
            const int p_km1 = lo + (int)(((long long)k * len) / 9);
            const int new_lo = k > 0 ? p_km1 + 1 : lo;
            const int new_hi = k < 8 ? p_k : hi;
            lo = new_lo; hi = new_hi;
        }
    }
    iA = __builtin_amdgcn_readlane(lo, 0);
    iB = __builtin_amdgcn_readlane(lo, 8);
    iC = __builtin_amdgcn_readlane(lo, 16);
    iD = __builtin_amdgcn_readlane(lo, 24);
    iF1 = __builtin_amdgcn_readlane(lo, 32);
    iF2 = __builtin_amdgcn_readlane(lo, 40);
    if (tB >= tC) { iB = iD; iC = iD; }
    iF1 = min(max(iF1, iB), iC);
    iF2 = min(max(iF2, iF1), iC);
    iN1 = min(max(__builtin_amdgcn_readlane(lo, 48), iF1), iF2);       // the bounds at FF_MID half-spans, inside [iF1, iF2]
    iN2 = min(max(__builtin_amdgcn_readlane(lo, 56), iN1), iF2);
}

// XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so each XCD gets a
// contiguous run of tiles: neighbouring tiles read almost the same line records.
__device__ __forceinline__ int xcd_tile(int b, int n_tiles, int natural = 0) {
    if (natural == 1) return b < n_tiles ? b : -1;
    if (natural >= 2) {
        // low-discrepancy order: step through the tiles with a stride near n_tiles/phi (made coprime
        // to n_tiles), so that at any time the resident workgroups sample the whole grid evenly and
        // the last ones to start are a random draw instead of one dense region
        if (b >= n_tiles) return -1;
        int stride = (int)(0.6180339887498949 * (double)n_tiles) | 1;
        auto gcd = [](int a, int c) { while (c) { const int t = a % c; a = c; c = t; } return a; };
        while (stride > 1 && gcd(stride, n_tiles) != 1) stride += 2;
        if (stride >= n_tiles) stride = 1;
        return (int)(((long long)b * stride) % n_tiles);
    }
    const int chunk = (n_tiles + 7) >> 3;
    const int slot = b >> 3;
    if (slot >= chunk) return -1;
    const int tile = (b & 7) * chunk + slot;
    return tile < n_tiles ? tile : -1;
}

template <int R>
__device__ __forceinline__ void store_points(double* __restrict__ out, int p0, int n_end, const double (&acc)[R]) {
    if (p0 + R <= n_end) {
        if (R >= 2) {
#pragma unroll
            for (int k = 0; k < R; k += 2) {
                double2 v; v.x = acc[k]; v.y = acc[k + (R >= 2 ? 1 : 0)];
                *reinterpret_cast<double2*>(out + p0 + k) = v;
            }
        } else {
            out[p0] = acc[0];
        }
    } else {
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (p0 + k < n_end) out[p0 + k] = acc[k];
    }
}

template <int R, int DIV, int GM>
__global__ __launch_bounds__(256) void xsec_accumulate_kernel(const AccumJob* __restrict__ jobs) {
    const AccumJob& J = jobs[blockIdx.y];
    const int tile = xcd_tile(blockIdx.x, J.n_tiles);
    if (tile < 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i32(threadIdx.x >> 6);
    const int n_end = J.p_end;          // this job computes work-grid points [p_begin, p_end)
    const long long wave_lo_ll = (long long)J.p_begin + (long long)tile * (256LL * R) + (long long)wave * (64LL * R);
    if (wave_lo_ll >= n_end) return;
    const int wlo = (int)wave_lo_ll;
    const int whi = min(wlo + 64 * R - 1, n_end - 1);
    const int H = J.H;
    int iA, iB, iC, iD;
    wave_line_ranges(J.cidx, J.n_lines, wlo, whi, H, lane, iA, iB, iC, iD);

    const int p0 = wlo + lane * R;
    const double x0 = (double)p0;
    const double Hf = (double)H;
    WaveAcc<R> S;
    S.init(J.flush_every);
    process_lines_scalar<R, true, DIV, GM>(J.hot, J.cold, iA, iB, x0, wlo, whi, Hf, S);
    process_lines_scalar<R, false, DIV, GM>(J.hot, J.cold, iB, iC, x0, wlo, whi, Hf, S);
    process_lines_scalar<R, true, DIV, GM>(J.hot, J.cold, iC, iD, x0, wlo, whi, Hf, S);
    S.flush();
    store_points<R>(J.out, p0, n_end, S.acc);
}

// ---- variant 3: wave-private LDS staging of the line records --------------------------------
// The scalar cache is not a streaming path: with one dependent s_load per (wave, line) the
// kernel ran at a tenth of the fp64 rate.  Here each wave streams its line range in chunks of
// 64 records with coalesced 16-byte vector loads (lane l fetches record c0+l), parks the chunk
// in its own 4 KB of LDS and reads every record back as a broadcast (all lanes, one address:
// conflict-free).  The next chunk's global loads are in flight while the current chunk is
// consumed, so neither HBM/L2 latency nor LDS latency is exposed, and no workgroup barrier is
// needed: waves stay independent.
//
// LS waves of a workgroup can share the same 64*R grid points and split that span's line range
// between them (small grids: more wavefronts without shrinking R); their partial sums meet in
// LDS in a fixed order, so results stay deterministic.
struct HotVals {
    double cf, a2, KL;
};

__device__ __forceinline__ HotVals lds_hot(const double* __restrict__ lh, int j) {
    const double* h = lh + j * 4;
    HotVals v;
    v.cf = h[0]; v.a2 = h[1]; v.KL = h[2];
    return v;
}

// Lorentz terms of records j0..j1-1 of the chunk parked in this wave's LDS, as a branch-free
// running-fraction loop: per point  d = d0+k, den = d*d+a2, t = K*D, N = N*den+t, D = D*den.
// The NEXT record's broadcast read is issued before the current record's arithmetic, so the LDS
// latency hides under the 5*R fp64 instructions (at the chunk end that read lands in the cold records behind slot 63 - at most
// `step` slots further, inside the wave's staging area - and is dropped: no clamp, the address just counts up).
// The flush test sits outside the inner loop (blocks of at most `every` lines), and Gaussian /
// plain-divide lines are handled after the chunk from bit masks, so the hot loop has no branch.
template <int R, bool MASKED>
__device__ __forceinline__ void rf_segment(const double* __restrict__ lh, int j0, int j1, double x0, double Hf,
                                           WaveAcc<R>& S, int step = 1, int phase = 0) {
    // records j0 <= j < j1 with j % step == phase (step = line split of the span, a power of two)
    int j = j0 + ((phase - j0) & (step - 1));
    while (j < j1) {
        const int left = (j1 - j + step - 1) / step;
        const int nb = min(S.every - S.cnt, left);
        HotVals nxt = lds_hot(lh, j);
#pragma unroll 2
        for (int t = 0; t < nb; ++t) {
            const HotVals cur = nxt;
            nxt = lds_hot(lh, j + (t + 1) * step);      // (past the chunk's last record: words of the cold records, read and never used)
            const double d0 = x0 - cur.cf;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double d = d0 + (double)k;
                const double den = fma(d, d, cur.a2);
                double K = cur.KL;
                if (MASKED) K = (fabs(d) <= Hf) ? K : 0.0;
                const double tt = K * S.D[k];
                S.N[k] = fma(S.N[k], den, tt);
                S.D[k] *= den;
            }
        }
        j += nb * step;
        S.cnt += nb;
        if (S.cnt >= S.every) S.flush();
    }
}

// Rare per-record work of a chunk, driven by the masks computed when the chunk was staged:
// gmask bit j: record j's Gaussian term can matter for some point of this wave;
// dmask bit j: record j's denominator is outside the running-fraction range -> plain divide
// (its LDS copy carries K = 0, a2 = 1 so the hot loop adds nothing for it).
// LDS slot of grid-point offset o within a wave's span (padded so that a lane writing its R
// consecutive points and a lane reading every 64th point are both nearly conflict-free)
__device__ __forceinline__ int span_slot(int o) { return o + (o >> 4); }

// Transposed Gaussian pass: FOUR records at once, lane group q = lane/16 takes one of them and lane t =
// lane%16 of the group walks the 16 consecutive points 16t .. 16t+15 of the span with the recurrence
// g(d+1) = g(d) r(d), r(d+1) = r(d) q2: two exp per 16 points instead of per 4 (25 instead of 51 wave
// instructions per record).  Sums go to G[16] per lane (point 16t+k of the span, partial over this
// group's records); gauss_runs_fold adds the four groups' partial sums to the owners of the points
// in a fixed order.  Lines flagged REC_LONG_RUN by K1 come here; MASKED for those whose support ends
// inside the span.
// RUN = 16 points per lane: 16 lanes per record, four records per pass (the description above).  RUN = 32 (round 6, the
// three-waves-per-SIMD build of the far-field kernel: -DLBL_FF_WPS=3 -DLBL_GAUSS_RUN=32): 8 lanes per record, eight records
// per pass, two exp per 32 points - 142 instead of 92 wave-instructions per pass for twice the records - and 64 instead of
// 32 registers of sums.
template <int RUN, bool MASKED>
__device__ __forceinline__ void gauss_runs(const double* __restrict__ lh, const double* __restrict__ lc,
                                           unsigned long long m, double xrun, double Hf, int lane, double (&G)[RUN]) {
    constexpr int LPR = 256 / RUN, NREC = 64 / LPR;       // lanes per record, records per pass
    const int q = lane / LPR;
    // The records of the mask as a list of bytes in the wave's LDS (behind the staged records), built once: every pass then
    // reads its four record numbers (one ds_read_u8 per lane) instead of scanning the mask with ~32 scalar instructions -
    // which are not free: each takes an issue slot of its wave (round 5: merged C3 K2 243.5 -> 238.7 us, bit-identical).
    unsigned char* list = reinterpret_cast<unsigned char*>(const_cast<double*>(lh) + 512);
    const int cnt = __popcll(m);
    __builtin_amdgcn_wave_barrier();
    if ((m >> lane) & 1ull)
        list[__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u))] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int j_first = __builtin_ctzll(m);
    for (int p = 0; p < cnt; p += NREC) {
        const int slot = p + q;
        const int j = slot < cnt ? (int)list[slot] : -1;
        const int jj = j < 0 ? j_first : j;
        const double cf = lh[jj * 4];
        const double* c = lc + jj * 4;
        const double KG = j < 0 ? 0.0 : c[0], b = c[1], q2 = c[2];
        const double d0 = xrun - cf;
        double g = KG * exp_clamped(-b * (d0 * d0));
        double rr = exp_clamped(fmin(-b * (2.0 * d0 + 1.0), 700.0));
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            if (MASKED) G[k] += (fabs(d0 + (double)k) <= Hf) ? g : 0.0;      // support ends inside the span (cls:394)
            else G[k] += g;
            g *= rr;
            rr *= q2;
        }
    }
}

// G (lane (q, t): points RUN t .. RUN t + RUN-1, partial sums of group q) -> acc (lane l: points R l .. R l + R-1), through
// the wave's LDS scratch.  RUN = 16: 2 KB, one group per round, groups added in the order 0..3.  RUN = 32: four regions of
// 272 doubles, four groups per round (group q in region q % 4), two rounds; groups added in the order 0..7.
template <int R, int RUN>
__device__ __forceinline__ void gauss_runs_fold(const double (&G)[RUN], double* scratch, int lane, double (&acc)[R]) {
    constexpr int LPR = 256 / RUN, NREC = 64 / LPR, NREG = RUN / 8, REGION = 272;      // regions in use at once: 2 (RUN 16: one suffices, kept at 1 below) / 4
    const int q = lane / LPR, t = lane % LPR;
    if (RUN == 16) {
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            __builtin_amdgcn_wave_barrier();
            if (q == round) {
#pragma unroll
                for (int k = 0; k < RUN; ++k) scratch[span_slot(16 * t + k)] = G[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < R; ++k) acc[k] += scratch[span_slot(lane * R + k)];
        }
    } else {
#pragma unroll
        for (int round = 0; round < NREC / NREG; ++round) {
            __builtin_amdgcn_wave_barrier();
            if (q / NREG == round) {
                double* reg = scratch + (q % NREG) * REGION;
#pragma unroll
                for (int k = 0; k < RUN; ++k) reg[span_slot(RUN * t + k)] = G[k];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < NREG; ++g) {
#pragma unroll
                for (int k = 0; k < R; ++k) acc[k] += scratch[g * REGION + span_slot(lane * R + k)];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

template <int R, int LONGG = 0, int RUN = 16>
__device__ __forceinline__ void chunk_extras(const double* __restrict__ lh, const double* __restrict__ lc,
                                             unsigned long long gmask, unsigned long long emask,
                                             unsigned long long dmask, unsigned long long imask, double x0, double Hf,
                                             WaveAcc<R>& S, unsigned long long lmask, double xrun, int lane,
                                             double (&G16)[RUN]) {
    // emask bit j: record j must use one exp per point (profile too narrow for the recurrence)
    // imask bit j: record j is an interior line (every point of the wave inside its support): no masking
    // lmask bit j: record j may take the transposed 16-point runs (LONGG kernels, R = 4)
    if (LONGG && R == 4) {
        // LONGG 1: interior lines only (far-field kernel: its spans see few masked Gaussian records, and the
        // masked instantiation costs it 1 % in registers and code); 2: masked ones too (all-direct kernel,
        // i.e. the narrow-window layers of a column, where most records end inside the span)
        const unsigned long long ml = gmask & ~emask & lmask & (LONGG >= 2 ? ~0ull : imask);
        if (ml & imask) gauss_runs<RUN, false>(lh, lc, ml & imask, xrun, Hf, lane, G16);
        if (LONGG >= 2 && (ml & ~imask)) gauss_runs<RUN, true>(lh, lc, ml & ~imask, xrun, Hf, lane, G16);
        gmask &= ~ml;
    }
    unsigned long long m = gmask & ~emask & imask;
    while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const double d0 = x0 - lh[j * 4];
        const double* c = lc + j * 4;
        gauss_term<R, 1, false>(c[0], c[1], true, d0, Hf, c[2], S.acc);
    }
    m = gmask & ~emask & ~imask;
    while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const double d0 = x0 - lh[j * 4];
        const double* c = lc + j * 4;
        gauss_term<R, 1, true>(c[0], c[1], true, d0, Hf, c[2], S.acc);
    }
    m = gmask & emask;
    while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const double d0 = x0 - lh[j * 4];
        const double* c = lc + j * 4;
        gauss_term<R, 0>(c[0], c[1], false, d0, Hf, 0.0, S.acc);
    }
    while (dmask) {
        const int j = __builtin_ctzll(dmask);
        dmask &= dmask - 1;
        const double d0 = x0 - lh[j * 4];
        const double* c = lc + j * 4;
        const double a2 = 1.0 / c[1], KL = c[3];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double t = KL / fma(d, d, a2);
            S.acc[k] += (fabs(d) <= Hf) ? t : 0.0;
        }
    }
}

// Lines of one job against the 64*R points of a wave (wlo..whi; x0 = this lane's first point):
// the 64-line chunks [c0, c0+64) for c0 = mA, mA+stride, ... below mD.  stride = 64 walks a
// contiguous range; stride = 64*LS deals the chunks of a span round-robin to the LS waves that
// share it, so each wave gets its share of the expensive lines near the span (Gaussian passes):
// with contiguous quarters the two middle waves did ~1.8x the work of the outer two.
// Lines below iB or from iC on end inside the wave's span and are masked per point.
// The wave streams the records in chunks of 64 through its own LDS (lh: hot halves, lc: cold
// halves), with the next chunk's loads in flight while the current one is consumed.
template <int R, int LONGG = 0, int RUN = 16>
__device__ __forceinline__ void accumulate_lines(const HotRec* hot, const ColdRec* cold, int mA, int mD, int iB, int iC,
                                                 int wlo, int whi, double x0, double Hf, double* lh, double* lc, int lane,
                                                 WaveAcc<R>& S, double (&G16)[RUN], int stride = 64, int step = 1, int phase = 0,
                                                 int nL = 0, int nR = 0x7fffffff) {
    // [nL, nR): the records whose LORENTZ term this walk adds (round 6: the lines between FF_MID and FF_FAR half-spans take
    // their Lorentz term from the series and only their Gaussian part from this walk)
    // step > 1 (with stride = 64): every wave of the span walks ALL chunks but takes only the records
    // j % step == phase of each, so the split is exact to a line.  Dealing whole chunks left one wave
    // of a two-way split with 128 of a span's ~210 near lines and the other with 82.
    const unsigned long long stripe = step == 1 ? ~0ull : step == 2 ? (0x5555555555555555ull << phase)
                                    : step == 4 ? (0x1111111111111111ull << phase) : (0x0101010101010101ull << phase);
    // global address space made explicit: a flat load would also count on lgkmcnt and every
    // LDS wait would then drain the prefetch of the next chunk
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1)))* GlobalF64x2;
    const GlobalF64x2 gh = (GlobalF64x2)(unsigned long long)hot;
    const GlobalF64x2 gc = (GlobalF64x2)(unsigned long long)cold;
    // software pipeline: registers hold the NEXT chunk's hot halves while LDS holds the current one
    v2f64 h0 = {0, 0}, h1 = {0, 0};
    if (mA + lane < mD) {
        const long long r = (long long)(mA + lane) * 2;
        h0 = gh[r]; h1 = gh[r + 1];
    }
    for (int c0 = mA; c0 < mD; c0 += stride) {
        const int c1 = min(c0 + 64, mD);
        __builtin_amdgcn_wave_barrier();
        // per-record branch decisions for the whole chunk, one lane per record
        const int ci = (int)h0.x;
        const int dgi = __double2loint(h1.y), fl = __double2hiint(h1.y);
        const bool valid = c0 + lane < c1;
        const bool mine = (stripe >> lane) & 1ull;
        const bool gauss = valid && mine && max(0, max(ci - whi, wlo - ci)) < dgi;
        const bool direct = valid && (fl & REC_DIRECT_DIV) != 0;
        const unsigned long long gmask = __ballot(gauss);
        const unsigned long long dmask = __ballot(direct && c0 + lane >= nL && c0 + lane < nR) & stripe;
        const unsigned long long emask = __ballot((fl & REC_NO_RECUR) != 0);
        const unsigned long long lmask = LONGG ? __ballot((fl & (RUN == 32 ? REC_LONG_RUN32 : REC_LONG_RUN)) != 0) : 0ull;
        v2f64 w0 = h0, w1 = h1;
        if (direct) { w0.y = 1.0; w1.x = 0.0; }          // a2 = 1, KL = 0 in the hot loop's copy
        reinterpret_cast<v2f64*>(lh)[lane * 2] = w0;
        reinterpret_cast<v2f64*>(lh)[lane * 2 + 1] = w1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // cold halves only for the records that will use them (about one in eight); they
        // land while the Lorentz loop below runs
        v2f64 c0v = {0, 0}, c1v = {0, 0};
        if (gauss || (direct && mine)) {
            const long long r = (long long)(c0 + lane) * 2;
            c0v = gc[r]; c1v = gc[r + 1];
        }
        if (c0 + stride + lane < mD && lane < 64) {
            const long long r = (long long)(c0 + stride + lane) * 2;
            h0 = gh[r]; h1 = gh[r + 1];
        }
        // the three classes of lines inside this chunk, as offsets into the chunk
        const int a1 = max(min(iB, c1), c0) - c0;
        const int b1 = max(min(iC, c1), c0) - c0;
        const int n0 = max(min(nL, c1), c0) - c0, n1 = max(min(nR, c1), c0) - c0;     // the chunk's share of [nL, nR)
        rf_segment<R, true>(lh, n0, min(a1, n1), x0, Hf, S, step, phase);
        rf_segment<R, false>(lh, max(a1, n0), min(b1, n1), x0, Hf, S, step, phase);
        rf_segment<R, true>(lh, max(b1, n0), n1, x0, Hf, S, step, phase);
        if (gmask | dmask) {
            if (gauss || (direct && mine)) {
                reinterpret_cast<v2f64*>(lc)[lane * 2] = c0v;
                reinterpret_cast<v2f64*>(lc)[lane * 2 + 1] = c1v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // records a1..b1-1 of the chunk are interior lines
            const unsigned long long imask = (b1 > a1) ? ((b1 - a1 >= 64 ? ~0ull : ((1ull << (b1 - a1)) - 1ull)) << a1) : 0ull;
            chunk_extras<R, LONGG, RUN>(lh, lc, gmask, emask, dmask, imask, x0, Hf, S, lmask,
                                        (double)(wlo + RUN * (lane % (256 / RUN))), lane, G16);
        }
    }
}

// ---- variant 5: far-field series for distant Lorentz lines -------------------------------------
// Most (line, span) pairs are far apart: with a +-5 cm^-1 window at 0.001 cm^-1 a line reaches
// 40 spans of 256 points and is more than 8 half-spans away from 75 % of them.  For such a line
// (centre c, delta = c - xc from the span centre xc, s = delta^2 + a^2) the Lorentz term is a
// smooth function of u = x - xc over the whole span and its generating function is that of the
// Chebyshev polynomials of the second kind:
//     K / ((u - delta)^2 + a^2) = (K/s) / (1 - 2 X t + t^2) = (K/s) sum_n U_n(X) t^n,
//     X = delta / sqrt(s),  t = u / sqrt(s),  |t| <= rho = h / |delta| <= 1 / FF_FAR.
// In the scaled variable tau = u/h the coefficients q_n = (K/s) U_n(X) (h/sqrt(s))^n obey
//     q_0 = K beta,  q_1 = alpha q_0,  q_{n+1} = alpha q_n - beta' q_{n-1},
//     beta = 1/s,  alpha = 2 h delta beta,  beta' = h^2 beta,
// i.e. three fp64 instructions per order and LINE instead of five per line and POINT.  One lane
// takes one line, adds its q_n to per-lane sums C_n; the 64 lanes' sums are then reduced once per
// span and the polynomial is evaluated at every lane's R points (Horner in tau, |tau| < 1).
// Truncation after FF_NT terms: |sum_{n>=NT} U_n t^n| <= rho^NT (NT (1-rho) + 1) / (1-rho)^2,
// relative to a term that is >= (K/s) / (1+rho)^2: with rho = 1/4 and NT = 30 that is 5.7e-17,
// below half an ulp (measured sweep of threshold/terms: 4/30 beat 8/20, 6/24, 3/40 and 12/16), so the series is as exact as the sum it replaces (it carries fewer roundings:
// measured 2e-16 against a long-double sum where the direct fp64 sum has 7e-16).
// Edge lines (support ends inside the span) and near lines keep the direct path; a far line whose
// Gaussian part still matters on the span (wide Doppler cores) gets its Gaussian pass as usual.
#ifndef LBL_FF_FAR
#define LBL_FF_FAR 4
#define LBL_FF_NT 30
#endif
constexpr int FF_FAR = LBL_FF_FAR;   // a line is far when |c - xc| >= FF_FAR * (32 R)
constexpr int FF_NT = LBL_FF_NT;     // series terms (exact mode: remainder below half an ulp)
// Budget mode (lbl_set_option "accuracy" 1: <= 1e-9 relative on the absorption coefficient instead of the last bits;
// BASELINE north_star asks for 1e-6).  The remainder after NT terms is at most
// rho^NT (NT (1 - rho) + 1) (1 + rho)^2 / (1 - rho)^2 of the line's own smallest term on the span; every term of the sum
// is positive, so the sum's relative error is below the worst line's.  At rho <= 1/4: 5.9e-10 with 18 terms, falling by
// 4x per further half-span.
constexpr int FF_NT_BUDGET = 18;
// (Measured and dropped: far from 3 half-spans on in budget mode - rho <= 1/3, 24 terms leave 2.4e-10 - takes a quarter of
// the near lines out of the direct loop but lengthens the reduction and the polynomial: C3 K2 231.9 vs 230.5 us, C2 55.1 vs
// 50.2, the column 3.78 vs 3.90 ms: no gain overall.  The threshold stays a template constant per mode.)
constexpr int FF_FAR_BUDGET = 4;
template <int NT> struct FarThreshold { static constexpr int value = NT == FF_NT_BUDGET ? FF_FAR_BUDGET : FF_FAR; };
// Round 6, the three-waves-per-SIMD build of the exact mode (32-point Gaussian runs, 168 VGPRs): the series takes the
// Lorentz terms from FF_MID = 3 half-spans on, with FF_NT_MID = 38 terms for the chunks nearer than 4 (rho <= 1/3: remainder
// <= rho^38 (38 (1 - rho) + 1) (1 + rho)^2 / (1 - rho)^2 = 7.8e-17 of a line's own smallest term on the span, below half an ulp).
// The lines between 3 and 4 half-spans - a quarter of what the running fraction used to walk - keep their Gaussian parts in
// the near walk's runs (the span table carries both bounds: slots 4, 5 at FF_FAR, slots 6, 7 at FF_MID).  Measured with the
// Gaussian parts switched off (diagnostic builds, same box): K2 165 -> 151.5 us on the merged 100-2500 cm^-1 cell.
constexpr int FF_MID = 3;
constexpr int FF_NT_MID = 38;
template <int NT> struct LorentzNear { static constexpr int value = NT == FF_NT_MID ? FF_MID : FarThreshold<NT>::value; };

template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // (bound_ctrl with full row and bank masks: a lane without a source reads 0 and the compiler needs no preset of the
    // destination - with bound_ctrl off every 64-bit move cost two extra v_mov_b32 0; round 6)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// sum over the 64 lanes, same value (and same summation tree) in every lane; all lanes must be active
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_move_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141>(v);      // row_half_mirror
    v += dpp_move_f64<0x140>(v);      // row_mirror: every lane of a row holds the row's sum
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Sums over the 64 lanes of NT per-lane values, eight at a time through the wave's LDS scratch
// (8 rows of 72 doubles): every lane parks its 8 values (row n, column lane, one pad per 8 lanes),
// lane (n, j) = (l >> 3, l & 7) adds the 8 entries of segment j of row n, three DPP steps add the
// 8 segments, and the row total is read back as a wave-uniform value.  About 50 instructions per
// 8 values instead of 23 per value for the all-lanes butterfly; fixed summation tree.
template <int NT>
__device__ __forceinline__ void wave_sum_rows(double (&C)[NT], double* scratch, int lane) {
    const int wr = lane + (lane >> 3);                       // column of this lane inside a row
    const int rd = (lane >> 3) * 72 + (lane & 7) * 9;        // first entry of this lane's segment
#pragma unroll
    for (int base = 0; base < NT; base += 8) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (base + n < NT) scratch[n * 72 + wr] = C[base + n];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double t = 0.0;
        if (base + (lane >> 3) < NT) {
            const double* seg = scratch + rd;
            t = ((seg[0] + seg[1]) + (seg[2] + seg[3])) + ((seg[4] + seg[5]) + (seg[6] + seg[7]));
        }
        t += dpp_move_f64<0xB1>(t);
        t += dpp_move_f64<0x4E>(t);
        t += dpp_move_f64<0x141>(t);
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (base + n < NT) C[base + n] = readlane_f64(t, n * 8);
    }
    __builtin_amdgcn_wave_barrier();
}

// Terms a chunk of far lines needs, by the distance d (in half-spans) of its NEAREST line from the span centre:
// rho <= 1/d, and the remainder bound above falls fast with d.  Exact mode (NT = 30, remainder below half an ulp of
// the line's own smallest term on the span): d >= 8: 20 terms (2.7e-17), d >= 16: 15 (1.7e-17), d >= 32: 12 (1.2e-17).
// Budget mode (NT = 18, 5.9e-10): d >= 8: 12 (2.8e-10), d >= 16: 9 (1.8e-10), d >= 32: 7 (2.6e-10).  With a window of
// +-39 half-spans two thirds of the far lines are beyond 16 half-spans: 17 terms on average instead of 30.
template <int NT> struct FarTerms {
    static constexpr bool exact = NT == FF_NT || NT == FF_NT_MID, budget = NT == FF_NT_BUDGET;
    static constexpr int t4 = budget ? 18 : (NT == FF_NT_MID ? FF_NT : NT);     // (from 4 half-spans on: 30 terms; nearer, NT = 38: the build whose series starts at 3)
    static constexpr int t8 = exact ? 20 : (budget ? 12 : NT);
    static constexpr int t16 = exact ? 15 : (budget ? 9 : NT);
    static constexpr int t32 = exact ? 12 : (budget ? 7 : NT);
};

// q_N0 .. q_{N1-1} of one line per lane added to the per-lane sums (qa, qb: q_{N0-2}, q_{N0-1}; carried on)
template <int N0, int N1, int NT>
__device__ __forceinline__ void series_terms(double al, double bp, double& qa, double& qb, double (&C)[NT]) {
#pragma unroll
    for (int n = N0; n < N1; ++n) {
        const double qn = fma(al, qb, -(bp * qa));
        C[n] += qn;
        // (an empty statement the scheduler may not move code across: without it the compiler runs the whole chain of
        // q_n first and sinks the additions behind the term-count branches - 20 more live values, spilled in the loop)
        if ((n & 1) == 1) asm volatile("" : "+v"(C[n]));
        qa = qb; qb = qn;
    }
}

// Series coefficients of the far lines m0, m0+stride*k.. (chunks of 64, one line per lane) below m1.
template <int R, int NT>
__device__ __forceinline__ void far_field_lines(const HotRec* hot, const ColdRec* cold, int m0, int m1, int stride,
                                                double xc, int wlo, int whi, double x0, double Hf, double* lh, double* lc,
                                                int lane, double (&C)[NT], WaveAcc<R>& S, int gx_lo = 0, int gx_hi = 0) {
    // records [gx_lo, gx_hi): their Gaussian parts belong to the near walk (lines between FF_MID and FF_FAR half-spans)
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1)))* GlobalF64x2;
    const GlobalF64x2 gh = (GlobalF64x2)(unsigned long long)hot;
    const GlobalF64x2 gc = (GlobalF64x2)(unsigned long long)cold;
    const double hh = 32.0 * R;
    v2f64 h0 = {0, 1}, h1 = {0, 0};
    if (m0 + lane < m1) {
        const long long r = (long long)(m0 + lane) * 2;
        h0 = gh[r]; h1 = gh[r + 1];
    }
    for (int c0 = m0; c0 < m1; c0 += stride) {
        const bool valid = c0 + lane < m1;
        const v2f64 w0 = h0, w1 = h1;
        if (c0 + stride + lane < m1) {
            const long long r = (long long)(c0 + stride + lane) * 2;
            h0 = gh[r]; h1 = gh[r + 1];
        }
        const int ci = (int)w0.x;
        const int dgi = __double2loint(w1.y), fl = __double2hiint(w1.y);
        const bool gauss = valid && max(0, max(ci - whi, wlo - ci)) < dgi && !(c0 + lane >= gx_lo && c0 + lane < gx_hi);
        const unsigned long long gmask = __ballot(gauss);
        if (gmask) {                                   // rare: a far line with a Gaussian part that reaches the span
            const unsigned long long emask = __ballot((fl & REC_NO_RECUR) != 0);
            v2f64 c0v = {0, 0}, c1v = {0, 0};
            if (gauss) {
                const long long r = (long long)(c0 + lane) * 2;
                c0v = gc[r]; c1v = gc[r + 1];
            }
            __builtin_amdgcn_wave_barrier();
            reinterpret_cast<v2f64*>(lh)[lane * 2] = w0;
            reinterpret_cast<v2f64*>(lh)[lane * 2 + 1] = w1;
            reinterpret_cast<v2f64*>(lc)[lane * 2] = c0v;
            reinterpret_cast<v2f64*>(lc)[lane * 2 + 1] = c1v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double unused[16];
            chunk_extras<R, 0>(lh, lc, gmask, emask, 0ull, ~0ull, x0, Hf, S, 0ull, 0.0, lane, unused);
            __builtin_amdgcn_wave_barrier();
        }
        const double dl = w0.x - xc;
        const double sq = fma(dl, dl, w0.y);
        double be = __builtin_amdgcn_rcp(sq);
        be = fma(fma(-sq, be, 1.0), be, be);
        be = fma(fma(-sq, be, 1.0), be, be);
        const double al = (dl * (2.0 * hh)) * be;
        const double bp = (hh * hh) * be;
        double qa = (valid ? w1.x : 0.0) * be;
        C[0] += qa;
        double qb = al * qa;
        C[1] += qb;
        // the chunk's nearest line (centres are sorted and a call stays on one side of the span: the first or the last
        // valid lane) decides how many terms the whole chunk takes; wave-uniform branches.  In integers (a centre is an
        // integer, xc = wlo + 32 R - 1/2): dmin2 = 2 |c - xc| from two v_readlane and scalar arithmetic instead of four
        // v_readlane and six fp64 vector instructions; the class limits are d half-spans = 32 R d points.
        const int nv = min(64, m1 - c0);
        const int ca = __builtin_amdgcn_readlane(ci, 0), cb = __builtin_amdgcn_readlane(ci, nv - 1);
        const int dmin2 = min(abs(2 * (ca - wlo) - (64 * R - 1)), abs(2 * (cb - wlo) - (64 * R - 1)));
        typedef FarTerms<NT> FT;
        series_terms<2, FT::t32, NT>(al, bp, qa, qb, C);
        if (dmin2 < 2 * 32 * 32 * R) {
            series_terms<FT::t32, FT::t16, NT>(al, bp, qa, qb, C);
            if (dmin2 < 2 * 16 * 32 * R) {
                series_terms<FT::t16, FT::t8, NT>(al, bp, qa, qb, C);
                if (dmin2 < 2 * 8 * 32 * R) {
                    series_terms<FT::t8, FT::t4, NT>(al, bp, qa, qb, C);
                    if (FT::t4 < NT && dmin2 < 2 * 4 * 32 * R) series_terms<FT::t4, NT, NT>(al, bp, qa, qb, C);
                }
            }
        }
    }
}

// ---- narrow windows: skewed line ranges ---------------------------------------------------------
// The upper layers of a column have windows of 50-640 points (pyradClasses.py:655: 5 P / 1013.25 cm^-1).
// In the span kernel above a wave walks every line that reaches ANY of its 64 R points, and most of
// those end inside the span: masked lanes (8 instead of 5 instructions per point, half of them for
// nothing), and a span may not be wider than a line's support, so the narrowest layers run R = 1.
// Here every LANE walks the lines that reach ITS R points: sorted centre indices make that a
// contiguous range [A', B') of records per lane, shifted by R points from lane to lane, so at
// iteration t lane l is on record A'(l) + t: the lanes run down the line list side by side, each at
// about the same offset from its current line (a skewed, systolic walk).  Every lane has the same
// number of lines up to density fluctuations, none of them masked except the few (0-2 per lane)
// that cover only part of the lane's R points: per lane  [A', A) partial, [A, B) full, [B, B') partial,
//     A' = #{c < p0 - H}   A = #{c < p0 + R-1 - H}   B = #{c <= p0 + H}   B' = #{c <= p0 + R-1 + H}.
// The four counts of all 64 lanes come from one scatter + scan per chunk: record j adds itself to the
// first threshold above its centre (ds_max of j+1: centres are sorted, so the count IS the largest
// such j+1), then a prefix maximum over the thresholds (R per lane, DPP scan across lanes).
// Records are staged per chunk of 112 in the wave's LDS (hot 32 B + cold 32 B) and read back with
// per-lane addresses (neighbouring lanes read the same or the next record: conflict-free).
// The Gaussian part of a line is evaluated by the lanes whose points it can still change (|d| < dgi, K1's
// cut-off), two exp per lane and line, in a walk of its own over per-lane ranges counted the same way, so
// that all lanes are at the cores of their current lines at the same time.  Summation order per point:
// Lorentz terms of the partially covering lines, then of the fully covering ones, ascending; then the
// Gaussian terms, ascending: fixed, so reruns are bit-identical.
// Records per chunk: what fits beside the counters so that four workgroups share a CU's 160 KB of LDS.
template <int R> struct SkewChunk { static constexpr int value = R >= 8 ? 80 : 112; };

__device__ __forceinline__ unsigned int dpp_max_u32(unsigned int v, unsigned int moved) { return v > moved ? v : moved; }

// inclusive prefix maximum over the 64 lanes
__device__ __forceinline__ unsigned int wave_prefix_max_u32(unsigned int v) {
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = dpp_max_u32(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}

// cnt[i] = #{records of the chunk with centre < base + i}, i = 0 .. 64 R; returns this lane's counts at
// i = R lane (lo) and i = R lane + R - 1 (hi).  cnt holds j+1 of the last record whose first counted
// threshold is i (0: none) when called.
template <int R>
__device__ __forceinline__ void skew_counts(const unsigned int* __restrict__ cnt, int lane, int& lo, int& hi) {
    unsigned int m[R];
#pragma unroll
    for (int k = 0; k < R; ++k) m[k] = cnt[lane * R + k];
#pragma unroll
    for (int k = 1; k < R; ++k) m[k] = m[k] > m[k - 1] ? m[k] : m[k - 1];
    const unsigned int inc = wave_prefix_max_u32(m[R - 1]);
    unsigned int ex = (unsigned int)__shfl_up((int)inc, 1, 64);
    if (lane == 0) ex = 0u;
    lo = (int)(ex > m[0] ? ex : m[0]);
    hi = (int)inc;
}

__device__ __forceinline__ int wave_max_i32(int v) {        // v >= 0; the result is wave-uniform
    return __builtin_amdgcn_readlane((int)wave_prefix_max_u32((unsigned int)v), 63);
}

// One lane's Lorentz walk over the records [ja, ja + la) and then (TWO) [jb, jb + lb) of the chunk (per-lane
// bounds), into the running fraction.  The trip count is the longest walk of the wave (one DPP maximum); a
// lane that has finished reads the chunk's sentinel record (K = 0, centre at the span start so that its
// denominators stay small).  The loop is branch-free (a lane-dependent `if` around the loop-carried sums
// costs dozens of register copies per iteration), in blocks of at most `every` iterations between flushes
// of the running fraction; the next record's LDS read is issued before the current record's arithmetic.
// `it` counts the wave's iterations since the last flush, so every lane's product of denominators stays bounded.
template <int R, bool MASKED, bool TWO>
__device__ __forceinline__ void skew_lorentz(const double* __restrict__ lh, int sentinel, int ja, int la, int jb, int lb,
                                             double x0, double Hf, WaveAcc<R>& S, int& it) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    const v2f64* rec = reinterpret_cast<const v2f64*>(lh);
    la = max(la, 0);
    const int len = la + (TWO ? max(lb, 0) : 0);
    const int T = wave_max_i32(len);
    const int jb_off = jb - la;
    auto index = [&](int t) {
        if (TWO) return t < la ? ja + t : (t < len ? jb_off + t : sentinel);
        return t < la ? ja + t : sentinel;
    };
    for (int t0 = 0; t0 < T;) {
        const int nb = min(S.every - it, T - t0);
        int jj = index(t0);
        v2f64 n0 = rec[jj * 2], n1 = rec[jj * 2 + 1];
#pragma unroll 2
        for (int t = 0; t < nb; ++t) {
            const v2f64 c0 = n0, c1 = n1;
            jj = index(t0 + t + 1);
            n0 = rec[jj * 2]; n1 = rec[jj * 2 + 1];
            const double d0 = x0 - c0.x;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double d = d0 + (double)k;
                const double den = fma(d, d, c0.y);
                double K = c1.x;
                if (MASKED) K = (fabs(d) <= Hf) ? K : 0.0;
                const double tt = K * S.D[k];
                S.N[k] = fma(S.N[k], den, tt);
                S.D[k] *= den;
            }
        }
        it += nb;
        t0 += nb;
        if (it >= S.every) { S.flush(); it = 0; }
    }
}

// Edge lines of a span of the far-field kernel (support ends inside the span) through the skewed walk: the
// left-edge lines [iA, iB) and the right-edge lines [iC, iD) are staged side by side (64 + 64 records per
// round), every lane counts the ones that reach its R points - a lane near the left end of the span has
// many left-edge and few right-edge lines, a lane near the right end the opposite, the sum is the same for
// all lanes - and walks them in one unmasked loop (lines covering all its points) and one masked loop (the
// 0-2 lines that cover only some).  14 unmasked iterations for all lanes instead of 28 masked ones.
// Rounds with a record whose Gaussian part reaches the span, or whose denominator needs the plain divide,
// take the masked span path (accumulate_lines); K1's cut-off makes that rare for windows this wide.
// (Rounds that need the masked span path are only MARKED here - bit r of `odd` for round r < 64, everything from round 64
// on wholesale - and run afterwards through one accumulate_lines call site of the kernel (edge_rounds_masked): inlined into
// every edge routine the rare path cost them registers and the kernel a third of its code size.)
template <int R>
__device__ __forceinline__ void skew_edges(const HotRec* hot, const ColdRec* cold, int iA, int iB, int iC, int iD, int wlo,
                                           int whi, int H, double x0, double Hf, double* lh, double* lc,
                                           unsigned int* cntL, unsigned int* cntR, int lane, WaveAcc<R>& S, unsigned long long& odd) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1)))* GlobalF64x2;
    const GlobalF64x2 gh = (GlobalF64x2)(unsigned long long)hot;
    constexpr int SENT = 128;                             // records 0..63: left-edge, 64..127: right-edge, 128: the sentinel
    auto slot = [&](int c, int base) { const int i = c - base + 1; return i < 0 ? 0 : (i > 64 * R + 1 ? 64 * R + 1 : i); };
    int round = 0;
    for (int cL = iA, cR = iC; (cL < iB || cR < iD) && round < 64; cL += 64, cR += 64, ++round) {
        const int nL = max(min(iB - cL, 64), 0), nR = max(min(iD - cR, 64), 0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < R; ++k) { cntL[lane * R + k] = 0u; cntR[lane * R + k] = 0u; }
        if (lane < 8) { cntL[64 * R + lane] = 0u; cntR[64 * R + lane] = 0u; }
        if (lane == 0) { lh[SENT * 4] = (double)wlo; lh[SENT * 4 + 1] = 1.0; lh[SENT * 4 + 2] = 0.0; lh[SENT * 4 + 3] = 0.0; }
        v2f64 l0 = {0, 1}, l1 = {0, 0}, r0 = {0, 1}, r1 = {0, 0};
        const bool vL = lane < nL, vR = lane < nR;
        if (vL) { const long long g = (long long)(cL + lane) * 2; l0 = gh[g]; l1 = gh[g + 1]; }
        if (vR) { const long long g = (long long)(cR + lane) * 2; r0 = gh[g]; r1 = gh[g + 1]; }
        const int ciL = (int)l0.x, ciR = (int)r0.x;
        // the walk adds Lorentz terms only: a record with a Gaussian part that reaches the span, or outside the
        // running-fraction range, sends this round through the masked span path
        const bool oddL = vL && (max(0, max(ciL - whi, wlo - ciL)) < __double2loint(l1.y) || (__double2hiint(l1.y) & REC_DIRECT_DIV));
        const bool oddR = vR && (max(0, max(ciR - whi, wlo - ciR)) < __double2loint(r1.y) || (__double2hiint(r1.y) & REC_DIRECT_DIV));
        if (__any(oddL || oddR)) { odd |= 1ull << round; continue; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (vL) {
            reinterpret_cast<v2f64*>(lh)[lane * 2] = l0;
            reinterpret_cast<v2f64*>(lh)[lane * 2 + 1] = l1;
            atomicMax(&cntL[slot(ciL, wlo - H)], (unsigned int)(lane + 1));              // A', A
        }
        if (vR) {
            reinterpret_cast<v2f64*>(lh)[(64 + lane) * 2] = r0;
            reinterpret_cast<v2f64*>(lh)[(64 + lane) * 2 + 1] = r1;
            atomicMax(&cntR[slot(ciR, wlo + H + 1)], (unsigned int)(lane + 1));          // B, B'
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int a_part, a_full, b_full, b_part;
        skew_counts<R>(cntL, lane, a_part, a_full);
        skew_counts<R>(cntR, lane, b_full, b_part);
        int it = S.cnt;
        // lines covering all of the lane's points: left-edge [a_full, nL), right-edge [0, b_full); then the partial ones
        skew_lorentz<R, false, true>(lh, SENT, a_full, nL - a_full, 64, b_full, x0, Hf, S, it);
        skew_lorentz<R, true, true>(lh, SENT, a_part, a_full - a_part, 64 + b_full, b_part - b_full, x0, Hf, S, it);
        S.cnt = it;
    }
}

// Round 5: the edge lines through the far-field series.  An edge line is 4-39 half-spans from the span centre, so over the
// points it covers its Lorentz term is the same smooth function the series of far_field_lines expands - but it covers only
// part of the span.  Left-edge lines are sorted by where their support ends, right-edge lines by where it starts: the lines
// that cover ALL R points of a lane are a SUFFIX of the left-edge list and a PREFIX of the right-edge list, at positions the
// threshold counts of the skewed walk already give (a_full, b_full).  So one lane takes one line, computes its series
// coefficients q_n (three instructions per term, as for a far line), a wave prefix sum over the lanes (left-edge lines loaded
// in reverse order, so that the prefix IS the suffix; all terms positive: no cancellation, fixed tree) turns q_n into "sum
// over the first j lines" for every j at once, and every lane picks the one partial sum that belongs to its points
// (ds_bpermute) and adds it to ITS coefficient C_n of the edge lines' polynomial, evaluated by Horner at its R points at
// the end.  ~52 instructions per term and 64 + 64 lines instead of 21 per line and lane: the 84 edge lines of a merged
// 100-2500 cm^-1 span cost ~900 wave-instructions instead of ~1,900 (14 % of the kernel, PMC).  The 0-2 lines per lane
// that cover only some of its points keep the masked walk; rounds with a Gaussian part that reaches the span or a
// denominator outside the running-fraction range keep the masked span path, as before.  Like the skewed walk it runs BEFORE
// the far lines' series phase, so that its NTE coefficients and that phase's NT are never live together.
// NTE terms: by the distance of the job's nearest edge line, H - 32 R points (FarTerms classes: 20 / 15 / 12 from 8 / 16 / 32
// half-spans on, exact to half an ulp); nearer than 8 half-spans 30 terms would cost what the walk costs: the walk stays.
// inclusive prefix sum over the 64 lanes (lanes without a source add 0); fixed summation tree.  The two steps across rows
// fetch with full masks and are switched on per row by a factor (m15: 1.0 in rows 1 and 3, m31: 1.0 in rows 2 and 3, else
// 0.0; x * 1.0 is exact and the values are finite, so the sums are those of masked moves) - a DPP move under a partial row
// mask needs its destination preset, two more instructions per step.
__device__ __forceinline__ double wave_prefix_sum_f64(double v, double m15, double m31) {
    v += dpp_move_f64<0x111>(v);                // row_shr:1
    v += dpp_move_f64<0x112>(v);                // row_shr:2
    v += dpp_move_f64<0x114>(v);                // row_shr:4
    v += dpp_move_f64<0x118>(v);                // row_shr:8
    v = fma(dpp_move_f64<0x142>(v), m15, v);    // row_bcast:15 -> rows 1, 3
    v = fma(dpp_move_f64<0x143>(v), m31, v);    // row_bcast:31 -> rows 2, 3
    return v;
}

template <int R>
__device__ __forceinline__ void series_edges(const HotRec* hot, const ColdRec* cold, int iA, int iB, int iC, int iD, int wlo,
                                             int whi, int H, double x0, double Hf, double xc, int nte, double* lh, double* lc,
                                             unsigned int* cntL, unsigned int* cntR, int lane, WaveAcc<R>& S, unsigned long long& odd) {
    // Round 6: no coefficient array.  The polynomial of the edge lines is summed in ASCENDING powers as the terms come -
    // v_k += p_n tau_k^n with a running power per point - so a lane keeps 3 R doubles (tau, tau^n, v) instead of NTE
    // coefficients, both sides share one pass, and the term loop is a real loop with a run-time count (one body instead of
    // three unrolled instantiations).  |tau| < 1 and the p_n fall by at least 8x per term (edge lines are >= 8 half-spans
    // away), so the ascending sum is as accurate as Horner's.  With the coefficient array the kernel needed 162 VGPRs,
    // spilled 35 at its 128 and wrote the wave's sums to scratch in every round (round-5 verdict: 625 MB of stores on
    // the column where 250 MB are compulsory).
    double tau[R], v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { tau[k] = ((x0 + (double)k) - xc) * (1.0 / (32.0 * R)); v[k] = 0.0; }
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1)))* GlobalF64x2;
    const GlobalF64x2 gh = (GlobalF64x2)(unsigned long long)hot;
    constexpr int SENT = 128;                             // records 0..63: left-edge, 64..127: right-edge, 128: the sentinel
    constexpr double hh = 32.0 * R;
    const double m15 = (lane & 16) ? 1.0 : 0.0, m31 = (lane & 32) ? 1.0 : 0.0;
    auto slot = [&](int c, int base) { const int i = c - base + 1; return i < 0 ? 0 : (i > 64 * R + 1 ? 64 * R + 1 : i); };
    int round = 0;
    for (int cL = iA, cR = iC; (cL < iB || cR < iD) && round < 64; cL += 64, cR += 64, ++round) {
        const int nL = max(min(iB - cL, 64), 0), nR = max(min(iD - cR, 64), 0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < R; ++k) { cntL[lane * R + k] = 0u; cntR[lane * R + k] = 0u; }
        if (lane < 8) { cntL[64 * R + lane] = 0u; cntR[64 * R + lane] = 0u; }
        if (lane == 0) { lh[SENT * 4] = (double)wlo; lh[SENT * 4 + 1] = 1.0; lh[SENT * 4 + 2] = 0.0; lh[SENT * 4 + 3] = 0.0; }
        v2f64 l0 = {0, 1}, l1 = {0, 0}, r0 = {0, 1}, r1 = {0, 0};
        const bool vL = lane < nL, vR = lane < nR;
        const int oL = nL - 1 - lane;                    // left-edge lines in REVERSE order over the lanes: prefix sums = suffix sums
        if (vL) { const long long g = (long long)(cL + oL) * 2; l0 = gh[g]; l1 = gh[g + 1]; }
        if (vR) { const long long g = (long long)(cR + lane) * 2; r0 = gh[g]; r1 = gh[g + 1]; }
        const int ciL = (int)l0.x, ciR = (int)r0.x;
        const bool oddL = vL && (max(0, max(ciL - whi, wlo - ciL)) < __double2loint(l1.y) || (__double2hiint(l1.y) & REC_DIRECT_DIV));
        const bool oddR = vR && (max(0, max(ciR - whi, wlo - ciR)) < __double2loint(r1.y) || (__double2hiint(r1.y) & REC_DIRECT_DIV));
        if (__any(oddL || oddR)) { odd |= 1ull << round; continue; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (vL) {
            reinterpret_cast<v2f64*>(lh)[oL * 2] = l0;
            reinterpret_cast<v2f64*>(lh)[oL * 2 + 1] = l1;
            atomicMax(&cntL[slot(ciL, wlo - H)], (unsigned int)(oL + 1));                // A', A
        }
        if (vR) {
            reinterpret_cast<v2f64*>(lh)[(64 + lane) * 2] = r0;
            reinterpret_cast<v2f64*>(lh)[(64 + lane) * 2 + 1] = r1;
            atomicMax(&cntR[slot(ciR, wlo + H + 1)], (unsigned int)(lane + 1));          // B, B'
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int a_part, a_full, b_full, b_part;
        skew_counts<R>(cntL, lane, a_part, a_full);
        skew_counts<R>(cntR, lane, b_full, b_part);
        // lines covering all of the lane's points: left-edge [a_full, nL) = the lanes 0 .. nL-1-a_full of the reversed
        // order, right-edge [0, b_full) = the lanes 0 .. b_full-1
        const int gL = nL - 1 - a_full, gR = b_full - 1;
        const double hasL = gL >= 0 ? 1.0 : 0.0, hasR = gR >= 0 ? 1.0 : 0.0;
        const int srcL = max(gL, 0), srcR = max(gR, 0);
        double alL, bpL, qaL, qbL, alR, bpR, qaR, qbR;
        auto start = [&](double cf, double a2, double K, double& al, double& bp, double& qa, double& qb) {
            const double dl = cf - xc, sq = fma(dl, dl, a2);
            double be = __builtin_amdgcn_rcp(sq);
            be = fma(fma(-sq, be, 1.0), be, be);
            be = fma(fma(-sq, be, 1.0), be, be);
            al = (dl * (2.0 * hh)) * be; bp = (hh * hh) * be;
            qa = K * be; qb = al * qa;
        };
        start(l0.x, l0.y, vL ? l1.x : 0.0, alL, bpL, qaL, qbL);
        start(r0.x, r0.y, vR ? r1.x : 0.0, alR, bpR, qaR, qbR);
        // sum over the lines that cover this lane's points, of term n of their series: the prefix sums of both sides,
        // each lane picking its own (left first, then right: fixed order)
        auto pick = [&](double qL, double qR) {
            const double pL = __shfl(wave_prefix_sum_f64(qL, m15, m31), srcL, 64);
            const double pR = __shfl(wave_prefix_sum_f64(qR, m15, m31), srcR, 64);
            return fma(pR, hasR, pL * hasL);
        };
        double pw[R];
        {
            const double p0 = pick(qaL, qaR), p1 = pick(qbL, qbR);
#pragma unroll
            for (int k = 0; k < R; ++k) { v[k] += p0; pw[k] = tau[k]; v[k] = fma(p1, pw[k], v[k]); }
        }
        // q_n = al q_(n-1) - bp q_(n-2) written over q_(n-2): two terms per trip, so that the pair (q_(n-2), q_(n-1)) never moves
        auto term = [&](double& oldL, double newestL, double& oldR, double newestR) {
            oldL = fma(alL, newestL, -(bpL * oldL));
            oldR = fma(alR, newestR, -(bpR * oldR));
            const double p = pick(oldL, oldR);
#pragma unroll
            for (int k = 0; k < R; ++k) { pw[k] *= tau[k]; v[k] = fma(p, pw[k], v[k]); }
        };
        int n = 2;
#pragma clang loop unroll(disable)
        for (; n + 1 < nte; n += 2) {
            term(qaL, qbL, qaR, qbR);
            term(qbL, qaL, qbR, qaR);
        }
        if (n < nte) term(qaL, qbL, qaR, qbR);
        // the 0-2 lines per lane that cover only some of its points: the masked walk, folded in at once
        int it = 0;
        skew_lorentz<R, true, true>(lh, SENT, a_part, a_full - a_part, 64 + b_full, b_part - b_full, x0, Hf, S, it);
        S.flush();
    }
#pragma unroll
    for (int k = 0; k < R; ++k) S.acc[k] += v[k];
}

// Fused sweep of a layer with ONE line list (lbl_layer_step_dev): the arithmetic and operation order of
// layer_sweep_kernel applied in the accumulate kernel's output stage, where the cross section of a point
// is in a register:  xs_m = 0 + cross section (pyradClasses.py:566-571), k = 0 + xs_m * conc * P / 1E4 /
// kB / T (pyradClasses.py:583, 707-712).  Layers with several line lists keep the separate sweep launch:
// both ways of folding them into this kernel were built and measured slower (DESIGN.md "what did not help":
// one workgroup walking all line lists of its points, -6 %; the last workgroup of a tile folding it, -3 %).
__device__ __forceinline__ void fused_fold(const FusedSweep& A, const AccumJob& J, double xsec, double& xs_m, double& kk) {
#pragma clang fp contract(off)
    if (J.chain_flags & CHAIN_MOL_FIRST) xs_m = 0.0;
    xs_m += xsec;
    if (J.chain_flags & CHAIN_MOL_LAST) kk += A.budget ? xs_m * A.factor : abs_coef_term(xs_m, J.conc, A.P, A.T, A.rT);
}

// after the last line list: transmittance and outgoing radiance of the point
// (pyradClasses.py:716, 784-787; pyradPlanck.py:38-44)
__device__ __forceinline__ void fused_finish(const FusedSweep& A, long long j, double kk) {
#pragma clang fp contract(off)
    if (A.abs_coef) A.abs_coef[j] = kk;
    const double tr = A.budget ? exp_neg_budget(kk * A.depth) : exp(-kk * A.depth);
    if (A.trans) A.trans[j] = tr;
    if (A.I_out) {
        const double nu = linspace_at(j, A.n, A.start, A.stop, A.step);
        const double B = A.budget ? planck_budget(nu, A.pa, A.pbkT) : planck_wn(nu, A.T, A.rT, A.pa, A.pb);
        const double Iin = A.I_in ? A.I_in[j] : (A.budget ? planck_budget(nu, A.pa, A.pbk_surface)
                                                          : planck_wn(nu, A.surface_T, A.r_surface_T, A.pa, A.pb));
        const double transmitted = tr * Iin;
        const double emitted = (1.0 - tr) * B;
        A.I_out[j] = transmitted + emitted;
    }
}

// What leaves an accumulate job for grid point j: the sum times the job's exact output scale (1.0 for a line list's cross
// section; 2^e for a merged layer job, whose records K1 weighted by conc P / 1E4 / k / T / 2^e: the absorption coefficient),
// stored if the job has an output array, and the sweep of the point when it is fused in.
__device__ __forceinline__ void output_point(const AccumJob& J, double* __restrict__ out, long long j, double sum) {
    const double t = sum * J.out_scale;
    if (out) out[j] = t;
    if (J.fuse.on == 2) {
        // merged layer job (lbl_layer_merged_step_dev): t is the layer's absorption coefficient (pyradClasses.py:707-712)
        fused_finish(J.fuse, j, t);
    } else if (J.fuse.on) {
        // a layer with ONE line list: the sweep of a point right here, its cross section is in a register
        double xs_m = 0.0, kk = 0.0;
        fused_fold(J.fuse, J, t, xs_m, kk);
        fused_finish(J.fuse, j, kk);
    }
}


// The edge rounds the skewed walk / the edge series left out (a record whose Gaussian part reaches the span or whose
// denominator needs the plain divide; everything beyond 64 rounds): the masked span path, one call site.
template <int R>
__device__ __forceinline__ void edge_rounds_masked(const HotRec* hot, const ColdRec* cold, int iA, int iB, int iC, int iD, int wlo,
                                                   int whi, double x0, double Hf, double* lh, double* lc, int lane, WaveAcc<R>& S,
                                                   unsigned long long odd) {
    double unused[16];                                 // (no 16-point Gaussian runs on this path: 4-point passes into the sums)
    while (odd) {
        const int r = __builtin_ctzll(odd);
        odd &= odd - 1;
        // (one call over [lo, hi): left- and right-edge records alike are "masked" records for accumulate_lines)
        for (int side = 0; side < 2; ++side) {
            const int lo = (side ? iC : iA) + 64 * r, end = side ? iD : iB;
            const int hi = min(lo + 64, end);
            if (lo < hi) accumulate_lines<R, 0>(hot, cold, lo, hi, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, unused, 64, 1, 0);
        }
    }
    for (int side = 0; side < 2; ++side) {             // beyond 64 rounds of 64 lines per side: wholesale
        const int lo = (side ? iC : iA) + 64 * 64, end = side ? iD : iB;
        if (lo < end) accumulate_lines<R, 0>(hot, cold, lo, end, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, unused, 64, 1, 0);
    }
}

// GRUN: points per lane of a Gaussian run (gauss_runs).  16: 128 VGPRs, four waves per SIMD.  32 (round 6; the production
// shape R = 4, LS = 1 with the series only): half the exp per point in the Gaussian parts for 32 more registers of sums -
// 164 VGPRs, three waves per SIMD, 43 KB of LDS per workgroup.  Same box, merged 100-2500 cm^-1 cell: K2 242.9 -> 233.1 us
// (-4 %), half of it (a shard of 2) -4 %, budget mode -3.5 %, the column's wide layers +-0; a launch that fits ONE round of
// the chip's 4,096 wave slots loses (a per-list shard of 8, 3,516 waves: 44.6 -> 46.4 us - it needs a second round at three
// waves per SIMD), so the host picks 32 for launches of more than 4,096 waves (lbl_set_option "accum_gauss_run").
template <int R, int LS, int NT = 0, int GRUN = 16>                                      // NT: far-field series terms (0: every pair direct)
__global__ __launch_bounds__((LS > 4 ? 64 * LS : 256), (R >= 4 ? (GRUN == 32 ? 3 : 4) : 1))   // HIP: min waves per SIMD
void xsec_accumulate_lds_kernel(const AccumJob* __restrict__ jobs, const int2* __restrict__ worklist) {
    constexpr bool FF = NT > 0;
    constexpr int NW = LS > 4 ? LS : 4;              // wavefronts per workgroup (LS = 8: 512 threads)
    constexpr int PG = NW / LS;                      // point groups (64*R points each) per workgroup
    // per wave: hot records [0,256), cold records [256,512); after the line loop the same words
    // hold the wave's 64*R sums in point order (+ padding) for the coalesced store
    static_assert(GRUN == 16 || (GRUN == 32 && FF && R == 4 && LS == 1), "32-point Gaussian runs: production shape of the far-field kernel only");
    constexpr int STAGE_MIN = GRUN == 32 ? 4 * 272 : FF ? 576 : 520;        // FF: 8 x 72 doubles for wave_sum_rows; else 512 + the 64-byte record list of gauss_runs; 32-point runs: four fold regions
    constexpr int STAGE = (68 * R > STAGE_MIN) ? 68 * R : STAGE_MIN;
    __shared__ double s_stage[NW][STAGE];
    // edge lines through the skewed walk (skew_edges): the production shape only (unsplit spans of 256 points)
    constexpr bool EDGE_SKEW = FF && LS == 1 && R == 4;
    __shared__ unsigned int s_ecnt[EDGE_SKEW ? NW : 1][2][EDGE_SKEW ? 64 * R + 8 : 1];

    // worklist: (job, tile) pairs of the whole launch sorted by decreasing line count (longest
    // first), built once per (line lists, grid) on the host; without it blockIdx.y is the job
    int job = blockIdx.y, tile;
    if (worklist) {
        const int2 wk = worklist[blockIdx.x];
        job = wk.x; tile = wk.y;
    } else {
        tile = xcd_tile(blockIdx.x, jobs[job].n_tiles, jobs[job].pad);
    }
    const AccumJob& J = jobs[job];
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i32(threadIdx.x >> 6);
    const int grp = wave / LS, part = wave % LS;
    const int n_end = J.p_end;
    const long long wave_lo_ll = (long long)J.p_begin + (long long)(tile < 0 ? 0 : tile) * (64LL * R * PG) + (long long)grp * (64LL * R);
    const bool active = tile >= 0 && wave_lo_ll < n_end;
    if (LS == 1 && !active) return;                  // no workgroup barrier below when waves do not share points
    const int wlo = active ? (int)wave_lo_ll : 0;
    const int whi = active ? min(wlo + 64 * R - 1, n_end - 1) : 0;
    const int H = J.H;
    const int p0 = wlo + lane * R;
    const double x0 = (double)p0;
    const double Hf = (double)H;
    WaveAcc<R> S;
    S.init(J.flush_every);
    double* lh = s_stage[wave];
    double* lc = s_stage[wave] + 256;

    // line ranges of this span: tabulated by the host with the schedule, else searched here
    const int32_t* tab = nullptr;
    if (active && J.span_tab) tab = J.span_tab + (size_t)((wlo - J.p_begin) / (64 * R)) * 8;
    if (active && !FF) {
        int iA, iB, iC, iD;
        if (tab) { iA = uniform_i32(tab[0]); iB = uniform_i32(tab[1]); iC = uniform_i32(tab[2]); iD = uniform_i32(tab[3]); }
        else wave_line_ranges(J.cidx, J.n_lines, wlo, whi, H, lane, iA, iB, iC, iD);
        // this wave's share of the span's lines: every LS-th chunk of 64, starting at chunk `part`
        double G[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) G[k] = 0.0;
        accumulate_lines<R, 2>(J.hot, J.cold, iA + part * 64, iD, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64 * LS);
        if (R == 4) gauss_runs_fold<R, 16>(G, s_stage[wave], lane, S.acc);
        S.flush();
    }
    if (active && FF) {
        // interior lines at least FF_FAR half-spans from the span centre go through the series; the
        // chunks of every class are dealt round-robin to the LS waves, each class starting at a
        // different wave so that the short classes do not all land on wave 0
        constexpr int FAR = FarThreshold<NT>::value;
        constexpr int NEAR = LorentzNear<NT>::value;          // the series takes the Lorentz terms from here on (< FAR: round 6)
        static_assert(NEAR == FAR || EDGE_SKEW, "the mid class needs the near walk that knows the Lorentz sub-range");
        const long long fl = (long long)wlo + 32 * R - 1 - (long long)FAR * 32 * R;
        const long long fr = (long long)wlo + 32 * R + (long long)FAR * 32 * R;
        const double xc = (double)wlo + (32.0 * R - 0.5);
        int iA, iB, iC, iD, iF1, iF2, iN1, iN2;
        if (tab) {           // wave-uniform values: keep them in scalar registers
            iA = uniform_i32(tab[0]); iB = uniform_i32(tab[1]); iC = uniform_i32(tab[2]); iD = uniform_i32(tab[3]);
            iF1 = uniform_i32(tab[4]); iF2 = uniform_i32(tab[5]);
            iN1 = NEAR < FAR ? uniform_i32(tab[6]) : iF1; iN2 = NEAR < FAR ? uniform_i32(tab[7]) : iF2;
        }
        else {
            wave_line_ranges_far(J.cidx, J.n_lines, wlo, whi, H, fl, fr, lane, iA, iB, iC, iD, iF1, iF2, iN1, iN2,
                                 (long long)wlo + 32 * R - 1 - (long long)FF_MID * 32 * R, (long long)wlo + 32 * R + (long long)FF_MID * 32 * R);
            if (!(NEAR < FAR)) { iN1 = iF1; iN2 = iF2; }
        }
        if (LBL_ABLATE(J, 8)) { iA = iB; iD = iC; }                       // (diagnostic builds, timing only: no edge lines)
        const bool any_far = !LBL_ABLATE(J, 256) && (iN1 - iB) + (iC - iN2) > 0;      // (256: far lines dropped)
        // edge lines first (skewed walk), while neither the series coefficients nor the Gaussian run sums are live
        // (also on spans without any far line - windows just above the kernel's limit, grid ends: their interior lines
        // are all near, [iF1, iF2) = [iB, iC))
        const bool edges_done = EDGE_SKEW;
#ifndef LBL_EDGE_SERIES
#define LBL_EDGE_SERIES 1
#endif
        constexpr bool EDGE_SERIES = EDGE_SKEW && LBL_EDGE_SERIES;       // (0: the skewed walk of round 3 everywhere, for A/B builds)
        if (edges_done) {
            unsigned int* eL = s_ecnt[EDGE_SKEW ? wave : 0][0];
            unsigned int* eR = s_ecnt[EDGE_SKEW ? wave : 0][1];
            // the job's nearest edge line is H - 32 R points from the span centre (wave-uniform: the job's window)
            typedef FarTerms<FF ? NT : FF_NT> FT;
            const int dmin = H - 32 * R;
            unsigned long long odd = 0ull;
            // a round of the series costs ~52 wave-instructions per term + ~250 whatever it holds, the walk ~21 per line: the
            // series from 42 / 49 / 62 edge lines per span on (a merged 100-2500 cm^-1 span has 84, one of its line lists 28:
            // measured on the per-list step, series for everything: K2 +4.5 %)
            const int n_edge = (iB - iA) + (iD - iC);
            const int nte = dmin >= 32 * 32 * R ? FT::t32 : dmin >= 16 * 32 * R ? FT::t16 : FT::t8;
            if (EDGE_SERIES && dmin >= 8 * 32 * R && n_edge * 21 >= 52 * nte + 250)
                series_edges<R>(J.hot, J.cold, iA, iB, iC, iD, wlo, whi, H, x0, Hf, xc, nte, lh, lc, eL, eR, lane, S, odd);
            else
                skew_edges<R>(J.hot, J.cold, iA, iB, iC, iD, wlo, whi, H, x0, Hf, lh, lc, eL, eR, lane, S, odd);
            if (odd || iB - iA > 64 * 64 || iD - iC > 64 * 64)
                edge_rounds_masked<R>(J.hot, J.cold, iA, iB, iC, iD, wlo, whi, x0, Hf, lh, lc, lane, S, odd);
        }
        if (any_far) {
            // the running fraction of the edge lines is folded in first: N = 0, D = 1 are then constants through the series
            // phase instead of 16 live registers beside its 60 of coefficients (28 instructions per span)
            S.flush();
            constexpr int NTC = FF ? NT : 2;               // (the array of the instantiations without the series is never touched)
            double C[NTC];
#pragma unroll
            for (int n = 0; n < NTC; ++n) C[n] = 0.0;
            // (the Gaussian parts of the records [iF1, iF2) belong to the near walk below)
            far_field_lines<R, NTC>(J.hot, J.cold, iB + ((part + 1) % LS) * 64, iN1, 64 * LS, xc, wlo, whi, x0, Hf, lh, lc, lane, C, S, iF1, iF2);
            far_field_lines<R, NTC>(J.hot, J.cold, iN2 + ((part + 2) % LS) * 64, iC, 64 * LS, xc, wlo, whi, x0, Hf, lh, lc, lane, C, S, iF1, iF2);
            wave_sum_rows<NTC>(C, s_stage[wave], lane);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double tau = ((x0 + (double)k) - xc) * (1.0 / (32.0 * R));
                double v = C[NTC - 1];
#pragma unroll
                for (int n = NTC - 2; n >= 0; --n) v = fma(v, tau, C[n]);
                S.acc[k] += v;
            }
        }
        // the direct classes: left-edge, near and right-edge lines.  Without far lines (narrow window)
        // they are one contiguous run and go through the first stream alone (one prologue, not three).
        // (the LS waves of a span interleave these classes line by line; series chunks are dealt whole)
        // (Gaussian parts of interior lines: 16-point runs into G, folded into the sums once per span)
        double G[GRUN];
#pragma unroll
        for (int k = 0; k < GRUN; ++k) G[k] = 0.0;
        if (LBL_ABLATE(J, 512)) {                                            // (512: near lines dropped)
        } else if (edges_done) {
            accumulate_lines<R, 1, GRUN>(J.hot, J.cold, iF1, iF2, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64, LS, part, iN1, iN2);
        } else {
            accumulate_lines<R, 1, GRUN>(J.hot, J.cold, iA, any_far ? iB : iD, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64, LS, part);
            if (any_far) {
                accumulate_lines<R, 1, GRUN>(J.hot, J.cold, iF1, iF2, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64, LS, part);
                accumulate_lines<R, 1, GRUN>(J.hot, J.cold, iC, iD, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64, LS, part);
            }
        }
        if (R == 4) gauss_runs_fold<R, GRUN>(G, s_stage[wave], lane, S.acc);
        S.flush();
    }

    // Results leave through LDS so that every store instruction writes 512 contiguous bytes
    // (a lane owns R CONSECUTIVE points; storing them directly would touch 64 cache lines per
    // instruction).  With LS > 1 the LS partial sums of a span meet here in a fixed order.
    double* mine = s_stage[wave];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) mine[span_slot(lane * R + k)] = S.acc[k];
    if (LS > 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    if (active) {
        // the LS waves of a span share its output rows (and the fused sweep of their points); every
        // point is summed over the waves in the same order whichever wave stores it
        double* __restrict__ out = J.out;
        const int w0 = wave - part;                      // first wave of this span
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (LS > 1 && (i % LS) != part) continue;        // row i belongs to wave i % LS of the span
            const int o = i * 64 + lane;
            double t = s_stage[w0][span_slot(o)];
            for (int q = 1; q < LS; ++q) t += s_stage[w0 + q][span_slot(o)];
            if (wlo + o < n_end) output_point(J, out, wlo + o, t);
        }
    }
}

// One lane's walk over the records [g0, g1) whose Gaussian part may reach its R points (a per-lane range
// again, from the largest reach of the chunk; K1's own cut-off of every record decides inside): two exp per
// lane and line, all lanes at the cores of their current lines at the same time.  Branches are wave-uniform;
// a lane that does not need the term runs it with amplitude 0.  MASKED: points beyond the line's support are
// switched off (needed only when the largest reach of the chunk comes within R points of the support's end).
template <int R, bool MASKED>
__device__ __forceinline__ void skew_gauss(const double* __restrict__ lh, int cold_off, int sentinel, int g0, int g1,
                                           double x0, double Hf, WaveAcc<R>& S) {
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    const v2f64* rec = reinterpret_cast<const v2f64*>(lh);
    const int len = max(g1 - g0, 0);
    const int T = wave_max_i32(len);
    for (int t = 0; t < T; ++t) {
        const int jc = t < len ? g0 + t : sentinel;
        const v2f64 h0 = rec[jc * 2];
        const double gf = lh[(jc * 2 + 1) * 2 + 1];
        const v2f64 gg = rec[cold_off + jc * 2];                     // KG, b
        const double q2 = lh[(cold_off + jc * 2 + 1) * 2];
        const double d0 = x0 - h0.x;
        const bool need = fabs(d0 + 0.5 * (R - 1)) < gf;           // the sentinel's reach is 0
        const bool recur = q2 >= 0.0 && R >= 4;
        const bool need_r = need && recur, need_n = need && !recur;
        if (__any(need_r))
            gauss_term<R, 1, MASKED>(need_r ? gg.x : 0.0, need_r ? gg.y : 0.0, true, d0, Hf, need_r ? q2 : 1.0, S.acc);
        if (__any(need_n))
            gauss_term<R, 0, MASKED>(need_n ? gg.x : 0.0, need_n ? gg.y : 0.0, false, d0, Hf, 0.0, S.acc);
    }
}

// LS (round 5): the LS waves of a workgroup share ONE span of 64 R points and take every LS-th record of its lines; their
// partial sums meet in LDS in a fixed order.  For the merged layer jobs: three times the lines per point make a chunk of
// records cover a third of the positions, less than the span is wide, so that only part of the lanes has lines in it and
// the walk runs as long as the busiest lane of every chunk (measured on the column's 17 narrow layers: merged 1.58 ms
// against 1.44 with one job per line list); dealt over 4 waves a chunk covers the whole span again.
template <int R, int LS = 1>
__global__ __launch_bounds__(256, 4) void xsec_accumulate_skew_kernel(const AccumJob* __restrict__ jobs,
                                                                   const int2* __restrict__ worklist) {
    static_assert(LS == 1 || LS == 2 || LS == 4, "line split of the skewed-range kernel");
    constexpr int PG = 4 / LS;                       // spans per workgroup
    constexpr int SKEW_CH = SkewChunk<R>::value;
    constexpr int NROUND = (SKEW_CH + 63) / 64;
    constexpr int NCNT = 64 * R + 8;                 // thresholds 0 .. 64 R, one dump slot, padding
    constexpr int COLD = (SKEW_CH + 1) * 2;           // first cold record, in 16-byte units
    __shared__ double s_rec[4][(SKEW_CH + 1) * 8];   // per wave: hot records 0 .. CH (CH: the sentinel), then the cold records 0 .. CH
    __shared__ unsigned int s_cnt[4][2][NCNT];
    int job = blockIdx.y, tile;
    if (worklist) {
        const int2 wk = worklist[blockIdx.x];
        job = wk.x; tile = wk.y;
    } else {
        tile = xcd_tile(blockIdx.x, jobs[job].n_tiles, jobs[job].pad);
    }
    const AccumJob& J = jobs[job];
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i32(threadIdx.x >> 6);
    const int n_end = J.p_end;
    const int grp = wave / LS, part = wave % LS;
    const long long wave_lo_ll = (long long)J.p_begin + (long long)(tile < 0 ? 0 : tile) * (64LL * R * PG) + (long long)grp * (64LL * R);
    const bool active = tile >= 0 && wave_lo_ll < n_end;
    if (LS == 1 && !active) return;                  // waves are independent: no workgroup barrier below
    const int wlo = active ? (int)wave_lo_ll : 0;
    const int whi = active ? min(wlo + 64 * R - 1, n_end - 1) : 0;
    const int H = J.H;
    const int p0 = wlo + lane * R;
    const double x0 = (double)p0;
    const double Hf = (double)H;
    WaveAcc<R> S;
    S.init(J.flush_every);
    double* lh = s_rec[wave];
    double* lc = s_rec[wave] + COLD * 2;
    unsigned int* cntL = s_cnt[wave][0];
    unsigned int* cntR = s_cnt[wave][1];

    int iA = 0, iD = 0;
    if (!active) {
    } else if (J.span_tab) {
        const int32_t* tab = J.span_tab + (size_t)((wlo - J.p_begin) / (64 * R)) * 8;
        iA = uniform_i32(tab[0]); iD = uniform_i32(tab[3]);
    } else {
        int iB, iC;
        wave_line_ranges(J.cidx, J.n_lines, wlo, whi, H, lane, iA, iB, iC, iD);
    }
    // this wave's records: iA + part, iA + part + LS, ...
    const int n_mine = iD - iA > part ? (iD - iA - part + LS - 1) / LS : 0;
    auto record = [&](int s) { return (long long)(iA + part) + (long long)s * LS; };
    typedef double v2f64 __attribute__((ext_vector_type(2)));
    typedef const v2f64 __attribute__((address_space(1)))* GlobalF64x2;
    const GlobalF64x2 gh = (GlobalF64x2)(unsigned long long)J.hot;
    const GlobalF64x2 gc = (GlobalF64x2)(unsigned long long)J.cold;
    auto zero_counters = [&]() {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < R; ++k) { cntL[lane * R + k] = 0u; cntR[lane * R + k] = 0u; }
        if (lane < 8) { cntL[64 * R + lane] = 0u; cntR[64 * R + lane] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // every centre of [iA, iD) lies in [wlo - H, whi + H], so these differences fit an int
    auto slot = [&](int c, int base) { const int i = c - base + 1; return i < 0 ? 0 : (i > 64 * R + 1 ? 64 * R + 1 : i); };
    int it = 0;
    if (lane == 0) {                                      // the sentinel: centre at the span start, a2 = 1, K = 0, no Gaussian reach
        lh[SKEW_CH * 4] = (double)wlo; lh[SKEW_CH * 4 + 1] = 1.0; lh[SKEW_CH * 4 + 2] = 0.0; lh[SKEW_CH * 4 + 3] = 0.0;
        lc[SKEW_CH * 4] = 0.0; lc[SKEW_CH * 4 + 1] = 0.0; lc[SKEW_CH * 4 + 2] = 1.0; lc[SKEW_CH * 4 + 3] = 0.0;
    }
    for (int c0 = 0; c0 < n_mine; c0 += SKEW_CH) {
        const int n = min(SKEW_CH, n_mine - c0);
        zero_counters();
        unsigned long long dmask[NROUND];
        int cen[NROUND];
        bool live[NROUND];
        int reach = 0;
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            dmask[r] = 0ull; cen[r] = 0; live[r] = false;
            if (r * 64 >= n) continue;                    // (wave-uniform)
            const int s = r * 64 + lane;
            const bool valid = s < n;
            v2f64 h0 = {0, 1}, h1 = {0, 0}, q0 = {0, 0}, q1 = {0, 0};
            if (valid) {
                const long long g = record(c0 + s) * 2;
                h0 = gh[g]; h1 = gh[g + 1];
                q0 = gc[g]; q1 = gc[g + 1];
            }
            const int dgi = valid ? __double2loint(h1.y) : 0, fl = __double2hiint(h1.y);
            const bool direct = valid && (fl & REC_DIRECT_DIV) != 0;
            dmask[r] = __ballot(direct);
            if (direct) { h0.y = 1.0; h1.x = 0.0; }                           // a2 = 1, KL = 0 in the walk's copy
            // Gaussian reach as a double: the term can matter for a lane whose R points come within dgi of the centre
            h1.y = dgi > 0 ? (double)dgi + 0.5 * (R - 1) : 0.0;
            reach = max(reach, dgi);
            live[r] = valid;
            cen[r] = (int)h0.x;
            if (valid) {
                reinterpret_cast<v2f64*>(lh)[s * 2] = h0;
                reinterpret_cast<v2f64*>(lh)[s * 2 + 1] = h1;
                reinterpret_cast<v2f64*>(lc)[s * 2] = q0;
                reinterpret_cast<v2f64*>(lc)[s * 2 + 1] = q1;
                // left family: base wlo - H (A', A); right family: base wlo + H + 1 (B, B')
                atomicMax(&cntL[slot(cen[r], wlo - H)], (unsigned int)(s + 1));
                atomicMax(&cntR[slot(cen[r], wlo + H + 1)], (unsigned int)(s + 1));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int a_part, a_full, b_full, b_part;
        skew_counts<R>(cntL, lane, a_part, a_full);
        skew_counts<R>(cntR, lane, b_full, b_part);
        if (a_full >= b_full) { a_full = b_part; b_full = b_part; }         // support narrower than the lane's R points: all masked
        // Gaussian ranges from the largest reach of the chunk: records with centre in [p0 - g, p0 + R-1 + g], g = reach - 1
        int g_lo = 0, g_hi = 0;
        const int gmax = wave_max_i32(reach);
        if (gmax > 0 && !LBL_ABLATE(J, 1)) {
            const int g = gmax - 1;
            zero_counters();
#pragma unroll
            for (int r = 0; r < NROUND; ++r) {
                if (live[r]) {
                    atomicMax(&cntL[slot(cen[r], wlo - g)], (unsigned int)(r * 64 + lane + 1));
                    atomicMax(&cntR[slot(cen[r], wlo + g + 1)], (unsigned int)(r * 64 + lane + 1));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int unused;
            skew_counts<R>(cntL, lane, g_lo, unused);
            skew_counts<R>(cntR, lane, unused, g_hi);
            g_lo = max(g_lo, a_part); g_hi = min(g_hi, b_part);             // only lines whose support reaches the lane's points
        }
        if (!LBL_ABLATE(J, 2)) skew_lorentz<R, true, true>(lh, SKEW_CH, a_part, a_full - a_part, b_full, b_part - b_full, x0, Hf, S, it);
        if (!LBL_ABLATE(J, 4)) skew_lorentz<R, false, false>(lh, SKEW_CH, a_full, b_full - a_full, 0, 0, x0, Hf, S, it);
        // a record within reach g of a lane's block has |d| <= g + R - 1 at every point of the lane
        if (gmax + R - 2 <= H) skew_gauss<R, false>(lh, COLD, SKEW_CH, g_lo, g_hi, x0, Hf, S);
        else skew_gauss<R, true>(lh, COLD, SKEW_CH, g_lo, g_hi, x0, Hf, S);
        // lines whose denominator is outside the running-fraction range (K1: REC_DIRECT_DIV): plain divide
#pragma unroll
        for (int r = 0; r < NROUND; ++r) {
            unsigned long long dm = dmask[r];
            while (dm) {
                const int s = r * 64 + __builtin_ctzll(dm);
                dm &= dm - 1;
                const double d0 = x0 - lh[s * 4];
                const double* c = lc + s * 4;
                const double a2 = 1.0 / c[1], KL = c[3];
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    const double d = d0 + (double)k;
                    const double t = KL / fma(d, d, a2);
                    S.acc[k] += (fabs(d) <= Hf) ? t : 0.0;
                }
            }
        }
    }
    S.flush();

    // results leave through LDS so that every store instruction writes 512 contiguous bytes; with LS > 1 the LS partial
    // sums of a span meet here, added in wave order whichever wave stores the row
    double* mine = s_rec[wave];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < R; ++k) mine[span_slot(lane * R + k)] = S.acc[k];
    if (LS > 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    if (!active) return;
    double* __restrict__ out = J.out;
    const int w0 = wave - part;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (LS > 1 && (i % LS) != part) continue;        // row i belongs to wave i % LS of the span
        const int o = i * 64 + lane;
        double t = s_rec[w0][span_slot(o)];
        for (int q = 1; q < LS; ++q) t += s_rec[w0 + q][span_slot(o)];
        if (wlo + o < n_end) output_point(J, out, wlo + o, t);
    }
}

#ifdef LBL_DIAG      // variant 4 (measured: no faster than 3): diagnostic builds only (make EXTRA=-DLBL_DIAG)
// ---- variant 4: balanced single-round partition -----------------------------------------------
// Small grids do not fill the chip evenly with one workgroup per span: C2 has 1.5-3 rounds of
// workgroups of very different length (line density varies 7x) and ran with the VALU only 55-69 %
// busy.  Here the unit of work is a (span, line) pair.  P1 finds every span's line range, P2
// prefix-sums the counts, and K2b gives each of the W resident wavefronts exactly ceil(P/W)
// consecutive pairs: every wave does the same number of line iterations and they all finish
// together.  A wave's share may start and end inside a span; such partial sums go to a slab
// (at most two per wave) and P3 adds the slabs of a split span in wave order, so the result is
// still deterministic.  Spans covered by one wave are stored directly.
__device__ __forceinline__ int find_job(const AccumJob* __restrict__ jobs, int n_jobs, int g, int j0 = 0) {
    int j = j0;
    while (j + 1 < n_jobs && g >= jobs[j].span_first + jobs[j].n_spans) ++j;
    return j;
}

template <int R>
__global__ __launch_bounds__(256) void span_ranges_kernel(const AccumJob* __restrict__ jobs, int n_jobs, int total_spans,
                                                          SpanRec* __restrict__ spans, unsigned int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int g = uniform_i32(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= total_spans) return;
    const AccumJob& J = jobs[find_job(jobs, n_jobs, g)];
    const int wlo = J.p_begin + (g - J.span_first) * (64 * R);
    const int whi = min(wlo + 64 * R - 1, J.p_end - 1);
    int iA, iB, iC, iD;
    wave_line_ranges(J.cidx, J.n_lines, wlo, whi, J.H, lane, iA, iB, iC, iD);
    if (lane == 0) {
        SpanRec r; r.iA = iA; r.iB = iB; r.iC = iC; r.iD = iD;
        spans[g] = r;
        counts[g] = (unsigned int)(iD - iA);
    }
}

#endif

// exclusive prefix sum of n counts into prefix[0..n] (single workgroup of 1024 threads)
static int scan_blocks(int n) { const int tiles = (n + 1023) / 1024; return tiles < 1 ? 1 : (tiles > 64 ? 64 : tiles); }

// Exclusive prefix sums of n counts, prefix[n] = the total.  Round 6: up to 64 workgroups instead of one (a single block spent
// 48-143 us on the 30,472 / 79,696 tile costs of a re-windowed column: 78 strided loads per thread, twice).  Block b owns a
// segment of whole 1024-element tiles; it first adds up everything BEFORE its segment (the whole block, coalesced - the last
// block reads the array once: 320 KB from L2), then scans its own tiles with a wave prefix + 16 wave totals.  No block waits
// for another.
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}
__global__ __launch_bounds__(1024) void scan_counts_kernel(const unsigned int* __restrict__ counts, int n,
                                                           unsigned long long* __restrict__ prefix) {
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles = (n + 1023) / 1024;
    const int tiles_per_block = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const long long lo = (long long)blockIdx.x * tiles_per_block * 1024;
    const long long hi = min(lo + (long long)tiles_per_block * 1024, (long long)n);
    if (n == 0) { if (blockIdx.x == 0 && t == 0) prefix[0] = 0ull; return; }
    if (lo >= n) return;
    unsigned long long s = 0;
    for (long long i = t; i < lo; i += 1024) s += counts[i];
    s = wave_incl_scan_u64(s, lane);
    if (lane == 63) s_wave[wave] = s;
    __syncthreads();
    if (t == 0) {
        unsigned long long b = 0;
        for (int w = 0; w < 16; ++w) b += s_wave[w];
        s_base = b;
    }
    __syncthreads();
    unsigned long long base = s_base;
    for (long long tile = lo; tile < hi; tile += 1024) {
        const long long i = tile + t;
        const unsigned long long v = i < hi ? (unsigned long long)counts[i] : 0ull;
        const unsigned long long incl = wave_incl_scan_u64(v, lane);
        __syncthreads();                                   // (s_wave of the previous tile has been read by everybody)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, total = 0;
        for (int w = 0; w < 16; ++w) { const unsigned long long x = s_wave[w]; if (w < wave) before += x; total += x; }
        if (i < hi) prefix[i] = base + before + incl - v;
        base += total;
    }
    if (hi == n && t == 0) prefix[n] = base;
}

#ifdef LBL_DIAG
// first index i in [0, n] with prefix[i] > key (prefix non-decreasing): 16-ary search by the
// first 16 lanes' worth of probes replicated over the wave
__device__ __forceinline__ int upper_bound_u64(const unsigned long long* __restrict__ a, int n, unsigned long long key,
                                               int lane) {
    const int jj = lane & 15;
    int lo = 0, hi = n;                                   // answer in [lo, hi]; a[n] treated as > key if none
    while (hi > lo) {
        const long long len = (long long)hi - lo;
        const int pos = lo + (int)(((long long)(jj + 1) * len) / 17);
        const bool le = a[pos] <= key;                    // pos < hi <= n: always a valid slot of prefix[0..n]
        const int k = __popcll(__ballot(le) & 0xFFFFull);
        const int p_k = lo + (int)(((long long)(k + 1) * len) / 17);
        const int p_km1 = lo + (int)(((long long)k * len) / 17);
        const int new_lo = k > 0 ? p_km1 + 1 : lo;
        const int new_hi = k < 16 ? p_k : hi;
        lo = new_lo; hi = new_hi;
    }
    return lo;
}

__device__ __forceinline__ unsigned long long pairs_per_worker(unsigned long long P, int W) {
    unsigned long long q = (P + (unsigned long long)W - 1) / (unsigned long long)W;
    return q < 64ull ? 64ull : q;                         // at least one chunk of lines per wave
}

template <int R>
__global__ __launch_bounds__(256, (R >= 8 ? 4 : 1)) void xsec_accumulate_balanced_kernel(
        const AccumJob* __restrict__ jobs, int n_jobs, int total_spans, const SpanRec* __restrict__ spans,
        const unsigned long long* __restrict__ prefix, int W, double* __restrict__ slab) {
    constexpr int STAGE = (68 * R > 512) ? 68 * R : 512;
    __shared__ double s_stage[4][STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = uniform_i32(threadIdx.x >> 6);
    const int w = uniform_i32(blockIdx.x * 4 + wave);
    if (w >= W) return;
    const unsigned long long P = prefix[total_spans];
    const unsigned long long Q = pairs_per_worker(P, W);
    const unsigned long long start = (unsigned long long)w * Q;
    if (start >= P) return;
    const unsigned long long end = (start + Q < P) ? start + Q : P;
    double* lh = s_stage[wave];
    double* lc = s_stage[wave] + 256;

    int s = uniform_i32(upper_bound_u64(prefix, total_spans, start, lane)) - 1;      // prefix[s] <= start < prefix[s+1]
    int j = 0;
    unsigned long long pos = start;
    while (pos < end) {
        const unsigned long long p_lo = prefix[s], p_hi = prefix[s + 1];
        if (p_hi <= pos) { ++s; continue; }               // empty span, or the previous one was just finished
        const int n_s = (int)(p_hi - p_lo);
        const int off = (int)(pos - p_lo);
        const int piece = (int)((end - pos < (unsigned long long)(n_s - off)) ? (end - pos) : (unsigned long long)(n_s - off));
        j = find_job(jobs, n_jobs, s, j);
        const AccumJob& J = jobs[j];
        const SpanRec r = spans[s];
        const int wlo = J.p_begin + (s - J.span_first) * (64 * R);
        const int n_end = J.p_end;
        const int whi = min(wlo + 64 * R - 1, n_end - 1);
        const double x0 = (double)(wlo + lane * R);
        const double Hf = (double)J.H;
        WaveAcc<R> S;
        S.init(J.flush_every);
        double unused[16];
        accumulate_lines<R, 0>(J.hot, J.cold, r.iA + off, r.iA + off + piece, r.iB, r.iC, wlo, whi, x0, Hf, lh, lc, lane, S, unused);
        S.flush();
        // coalesced store through LDS: whole spans straight to the output, partial ones to the slab
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < R; ++k) lh[span_slot(lane * R + k)] = S.acc[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool whole = (off == 0) && (piece == n_s);
        double* __restrict__ dst = whole ? (J.out + wlo)
                                         : (slab + ((size_t)w * 2 + (pos == start ? 0 : 1)) * (size_t)(64 * R));
        const int limit = whole ? (n_end - wlo) : 64 * R;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int o = i * 64 + lane;
            if (o < limit) dst[o] = lh[span_slot(o)];
        }
        __builtin_amdgcn_wave_barrier();
        pos += (unsigned long long)piece;
    }
}

// P3: one wave per span: zeros for spans no line reaches, slab sums (in wave order) for split spans
template <int R>
__global__ __launch_bounds__(256) void span_finalize_kernel(const AccumJob* __restrict__ jobs, int n_jobs, int total_spans,
                                                            const unsigned long long* __restrict__ prefix, int W,
                                                            const double* __restrict__ slab) {
    const int lane = threadIdx.x & 63;
    const int g = uniform_i32(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= total_spans) return;
    const AccumJob& J = jobs[find_job(jobs, n_jobs, g)];
    const int wlo = J.p_begin + (g - J.span_first) * (64 * R);
    const int n_end = J.p_end;
    double* __restrict__ out = J.out;
    const unsigned long long p_lo = prefix[g], p_hi = prefix[g + 1];
    if (p_hi == p_lo) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int o = i * 64 + lane;
            if (wlo + o < n_end) out[wlo + o] = 0.0;
        }
        return;
    }
    const unsigned long long Q = pairs_per_worker(prefix[total_spans], W);
    const int w0 = (int)(p_lo / Q), w1 = (int)((p_hi - 1) / Q);
    if (w0 == w1) return;                                  // stored directly by its only wave
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int o = i * 64 + lane;
        double t = 0.0;
        for (int w = w0; w <= w1; ++w) {
            // a wave that started before this span contributes its LAST piece (slot 1); a wave that
            // starts inside (or at) the span contributes its FIRST piece (slot 0)
            const int slot = (w == w0 && (unsigned long long)w0 * Q < p_lo) ? 1 : 0;
            t += slab[((size_t)w * 2 + slot) * (size_t)(64 * R) + o];
        }
        if (wlo + o < n_end) out[wlo + o] = t;
    }
}

// Number of wavefronts that are resident at once: the partition must fit in ONE round, so ask the
// runtime how many 256-thread workgroups of the kernel a CU holds (registers, LDS) and stay one
// workgroup per CU below the answer when it is at the 8-block edge (the API over-reports there for
// SGPR-heavy kernels, MI355X_MICROARCH.md "Residency and cooperative launch").
template <int R>
static int balanced_workers_r(int n_cu) {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, xsec_accumulate_balanced_kernel<R>, 256, 0) != hipSuccess || blocks < 1)
        blocks = 2;
    if (blocks >= 8) blocks = 7;
    return (n_cu > 0 ? n_cu : 256) * blocks * 4;
}

int balanced_workers(int R, int n_cu) {
    switch (R) {
        case 1: return balanced_workers_r<1>(n_cu);
        case 2: return balanced_workers_r<2>(n_cu);
        case 4: return balanced_workers_r<4>(n_cu);
        default: return balanced_workers_r<8>(n_cu);
    }
}

template <int R>
static void launch_balanced_r(const AccumJob* d_jobs, int n_jobs, int total_spans, int n_workers, SpanRec* spans,
                              unsigned int* counts, unsigned long long* prefix, double* slab, hipStream_t s) {
    const int span_blocks = (total_spans + 3) / 4;
    hipLaunchKernelGGL((span_ranges_kernel<R>), dim3(span_blocks), dim3(256), 0, s, d_jobs, n_jobs, total_spans, spans, counts);
    hipLaunchKernelGGL(scan_counts_kernel, dim3(scan_blocks(total_spans)), dim3(1024), 0, s, counts, total_spans, prefix);
    hipLaunchKernelGGL((xsec_accumulate_balanced_kernel<R>), dim3((n_workers + 3) / 4), dim3(256), 0, s, d_jobs, n_jobs,
                       total_spans, spans, prefix, n_workers, slab);
    hipLaunchKernelGGL((span_finalize_kernel<R>), dim3(span_blocks), dim3(256), 0, s, d_jobs, n_jobs, total_spans, prefix,
                       n_workers, slab);
}

void launch_accumulate_balanced(const AccumJob* d_jobs, int n_jobs, int total_spans, int R, int n_workers,
                                SpanRec* spans, unsigned int* counts, unsigned long long* prefix, double* slab,
                                hipStream_t s) {
    if (n_jobs <= 0 || total_spans <= 0) return;
    switch (R) {
        case 1: launch_balanced_r<1>(d_jobs, n_jobs, total_spans, n_workers, spans, counts, prefix, slab, s); break;
        case 2: launch_balanced_r<2>(d_jobs, n_jobs, total_spans, n_workers, spans, counts, prefix, slab, s); break;
        case 4: launch_balanced_r<4>(d_jobs, n_jobs, total_spans, n_workers, spans, counts, prefix, slab, s); break;
        default: launch_balanced_r<8>(d_jobs, n_jobs, total_spans, n_workers, spans, counts, prefix, slab, s); break;
    }
}

#else
// (the production library carries no balanced kernel: lbl_set_option refuses accum_variant 4)
int balanced_workers(int, int) { return 0; }
void launch_accumulate_balanced(const AccumJob*, int, int, int, int, SpanRec*, unsigned int*, unsigned long long*, double*, hipStream_t) {}
#endif

// ----------------------------------------------------------------------------------------
// Schedule of a launch group, built on the device (time to first spectrum: a pressure or range change
// re-windows the layer, pyradClasses.py:734-752 -> resetData cls:45-56, and the next getter recomputes)
// ----------------------------------------------------------------------------------------
// What a launch group's accumulate kernels need besides the line records: for every span of 64 R points the six
// lower bounds of its edge / near / far lines in the sorted centre indices (one 32-byte row of the span table), and
// the dispatch order of its (job, tile) workgroups - longest first, XCD-partitioned when the launch has several
// rounds, bin-packed per CU when it has one (group_schedule in lbl_api.hip has the reasoning and the measurements).
// Until round 4 the host built both from its own evaluation of the centre indices: 4 ms for the 100-2500 cm^-1
// cell and 80 ms for the 30-layer column on one host thread, per re-windowing.  Here the bounds are searched in the
// very array K1 wrote (cidx), right after K1 in the same stream, and the order is sorted on the chip; nothing is
// copied back and the host never waits.  Dispatch order never changes a result; the span tables are the lower
// bounds of the same integers either way (tests/test_gpu_parity.py::test_device_schedule_*).
__global__ __launch_bounds__(256) void sched_spans_kernel(const SchedJob* __restrict__ jobs, int n_jobs, int total_spans, int R,
                                                          int spans_per_tile, long long far_reach, double cost_near,
                                                          double cost_edge, double cost_far, double cost_fixed,
                                                          int32_t* __restrict__ tabs, unsigned int* __restrict__ tile_cost,
                                                          int2* __restrict__ items) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;               // span of the group, job-major
    if (g >= total_spans) return;
    int k = 0;
    while (k + 1 < n_jobs && g >= jobs[k + 1].span_first) ++k;
    const SchedJob J = jobs[k];
    const int q = g - J.span_first;                                     // span of the job's shard
    const long long span = 64LL * R;
    const long long lo = (long long)J.p_begin + (long long)q * span;
    const long long hi = min(lo + span - 1, (long long)J.p_end - 1);
    const long long H = J.H;
    int iA = lower_bound_i32(J.cidx, J.n_lines, lo - H);
    int iB = lower_bound_i32(J.cidx, J.n_lines, hi - H);
    int iC = lower_bound_i32(J.cidx, J.n_lines, lo + H + 1);
    int iD = lower_bound_i32(J.cidx, J.n_lines, hi + H + 1);
    if (hi - H >= lo + H + 1) { iB = iD; iC = iD; }                     // span wider than the support: no interior line
    int iF1 = iB, iF2 = iC;
    if (far_reach > 0) {                                                // (same arithmetic as wave_line_ranges_far)
        iF1 = min(max(lower_bound_i32(J.cidx, J.n_lines, lo + 32 * R - far_reach), iB), iC);     // first line with c > fl
        iF2 = min(max(lower_bound_i32(J.cidx, J.n_lines, lo + 32 * R + far_reach), iF1), iC);    // first line with c >= fr
    }
    int iN1 = iF1, iN2 = iF2;                                           // the bounds at FF_MID half-spans, inside [iF1, iF2]
    if (far_reach > 0) {
        const long long mid_reach = (long long)FF_MID * 32 * R;
        iN1 = min(max(lower_bound_i32(J.cidx, J.n_lines, lo + 32 * R - mid_reach), iF1), iF2);
        iN2 = min(max(lower_bound_i32(J.cidx, J.n_lines, lo + 32 * R + mid_reach), iN1), iF2);
    }
    int32_t* e = tabs + ((size_t)J.span_first + (size_t)q) * 8;
    e[0] = iA; e[1] = iB; e[2] = iC; e[3] = iD; e[4] = iF1; e[5] = iF2; e[6] = iN1; e[7] = iN2;
    // wave-instructions of the span, as group_schedule prices them
    const double n_far = (double)((iF1 - iB) + (iC - iF2)), n_edge = (double)((iB - iA) + (iD - iC)), n_near = (double)(iF2 - iF1);
    const double c = n_near * cost_near + n_edge * cost_edge + n_far * cost_far + cost_fixed;
    const int tile = q / spans_per_tile;
    atomicAdd(&tile_cost[J.tile_first + tile], (unsigned int)(c + 0.5));          // integer adds: any order, same sum
    if (q % spans_per_tile == 0) { int2 it; it.x = k; it.y = tile; items[J.tile_first + tile] = it; }
}

// ascending bitonic sort of n (a power of two) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            // one compare-exchange per thread and pass: pair p = (i, i | j) with bit j of i clear, so every lane works (walking
            // all i and skipping the upper partners left half of every wave idle: 0.59 ms for the 16,384 keys of an XCD's part
            // of the column's narrow layers)
            for (int p = threadIdx.x; p < (n >> 1); p += blockDim.x) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int l = i | j;
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
        }
    }
    __syncthreads();
}

// key that sorts by decreasing cost, ties by increasing position (what std::stable_sort gives the host)
__device__ __forceinline__ unsigned long long sched_key(unsigned int cost, int pos) {
    return ((unsigned long long)(0xFFFFFFFFu - cost) << 32) | (unsigned int)pos;
}

// chunk (of `chunks` equal shares of the total cost, in positional order) that item i falls into
__device__ __forceinline__ int sched_chunk_of(const unsigned long long* __restrict__ prefix, int i, double share, int chunks) {
    const int c = (int)((double)prefix[i] / share);
    return c < chunks ? c : chunks - 1;
}

// Launches of several rounds: XCD-partitioned longest-first.  Workgroup i of the accumulate launch runs on XCD i mod 8
// and every XCD has its own L2: the positional tile sequence is cut into `chunks` (8 x 32) pieces of equal cost,
// piece c goes to part c mod 8, every part is sorted longest-first and the parts are interleaved - XCD x reads the
// records of part x only.
// Parts hold different numbers of items: up to the smallest part the interleave is strict (rank r of part x at slot
// 8 r + x); what the longer parts have left follows round by round over the parts that still have items (their
// cheapest tiles; the XCD alignment of that tail does not matter).
// Round 6: three kernels over the whole chip instead of one workgroup per part.  (1) sched_parts_kernel writes every part's keys,
// in positional order, to global scratch; (2) sched_tile_sort_kernel sorts every tile of 1,024 keys by itself; (3)
// sched_rank_xcd_kernel, one thread per key, adds up how many keys of the part's OTHER tiles sort before its own (a binary search
// per tile) - keys are unique (they end in the position), so own place + those counts IS the key's place in the sorted part -
// and writes its item straight to the dispatch list.  The 79,696 tiles of a re-windowed column's narrow layers took the bitonic
// sort in LDS (one workgroup per part, 16,384 keys, 105 passes) 0.36 ms.  Same order as that sort (and as std::stable_sort on
// the host).
__global__ __launch_bounds__(1024) void sched_parts_kernel(const unsigned long long* __restrict__ prefix,
                                                           const unsigned int* __restrict__ tile_cost, int N, int chunks,
                                                           unsigned long long* __restrict__ g_keys, int g_stride,
                                                           int* __restrict__ part_count) {
    __shared__ int s_start[8 * 64 + 1];               // first item of every chunk (chunks <= 512)
    const int x = blockIdx.x;
    const double share = (double)prefix[N] / (double)chunks + 1e-9;
    for (int c = threadIdx.x; c <= chunks; c += blockDim.x) {
        // first i in [0, N] whose chunk is >= c (chunk numbers do not decrease with i)
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sched_chunk_of(prefix, mid, share, chunks) < c) lo = mid + 1; else hi = mid;
        }
        s_start[c] = (c == chunks) ? N : lo;
    }
    __syncthreads();
    unsigned long long* keys = g_keys + (size_t)x * (size_t)g_stride;
    int off = 0;
    for (int c = x; c < chunks; c += 8) {
        const int a = s_start[c], b = s_start[c + 1];
        for (int i = a + (int)threadIdx.x; i < b; i += blockDim.x) keys[off + (i - a)] = sched_key(tile_cost[i], i);
        off += b - a;
    }
    if (threadIdx.x == 0) part_count[x] = off;
}

// (2) every tile of 1,024 keys of a part sorted by itself (bitonic, in LDS; the last tile padded with keys that sort last)
__global__ __launch_bounds__(256) void sched_tile_sort_kernel(unsigned long long* __restrict__ g_keys, int g_stride,
                                                              const int* __restrict__ part_count, int max_part) {
    __shared__ unsigned long long s_keys[1024];
    const int x = blockIdx.y;
    const int n_x = part_count[x];
    const int t0 = blockIdx.x * 1024;
    if (n_x > max_part || t0 >= n_x) return;                  // (whole workgroup)
    unsigned long long* keys = g_keys + (size_t)x * (size_t)g_stride + t0;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_keys[i] = t0 + i < n_x ? keys[i] : ~0ull;
    bitonic_sort_lds(s_keys, 1024);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) keys[i] = s_keys[i];
}

// (3) one thread per key: its place in its own sorted tile + the number of smaller keys in every other tile of the part (a
// binary search in the tile staged in LDS) is its place in the sorted part; the item goes straight to the dispatch list.
__global__ __launch_bounds__(1024) void sched_rank_xcd_kernel(const unsigned long long* __restrict__ g_keys, int g_stride,
                                                              const int* __restrict__ part_count, const int2* __restrict__ items,
                                                              int2* __restrict__ worklist, int max_part) {
    __shared__ unsigned long long s_tile[1024];
    const int x = blockIdx.y;
    const int n_x = part_count[x];
    const int t_mine = blockIdx.x;
    if (n_x > max_part || t_mine * 1024 >= n_x) return;         // (whole workgroup: no barrier is left behind; larger parts: the sort below)
    const unsigned long long* keys = g_keys + (size_t)x * (size_t)g_stride;
    const unsigned long long key = keys[t_mine * 1024 + threadIdx.x];           // (padding: ~0, sorted behind the tile's keys)
    int rank = threadIdx.x;
    const int n_tiles = (n_x + 1023) / 1024;
    for (int t = 0; t < n_tiles; ++t) {
        if (t == t_mine) continue;                            // (uniform over the workgroup)
        __syncthreads();
        s_tile[threadIdx.x] = keys[t * 1024 + threadIdx.x];
        __syncthreads();
        int lo = 0, hi = 1024;                                // first index whose key is not below mine (keys are unique): 0 .. 1024
        while (lo < hi) {                                     // (11 steps at most)
            const int mid = (lo + hi) >> 1;
            if (s_tile[mid] < key) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (key == ~0ull) return;
    int cnt[8], m = part_count[0];
#pragma unroll
    for (int y = 0; y < 8; ++y) { cnt[y] = part_count[y]; m = min(m, cnt[y]); }
    const int r = rank;
    const int src = (int)(unsigned int)(key & 0xFFFFFFFFull);
    long long pos;
    if (r < m) {
        pos = 8LL * r + x;
    } else {
        pos = 8LL * m;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            pos += max(0, min(cnt[y], r) - m);                     // full rounds m .. r-1
            if (y < x && cnt[y] > r) pos += 1;                      // parts ahead of x in round r
        }
    }
    worklist[pos] = items[src];
}

// Parts of more than kRankPartMax items (one part can hold most of a group's cheap tiles when a few tiles carry most of the
// cost; the build covers up to 2^20 tiles) are not ranked - quadratic - but sorted: one workgroup per such part, bitonic, in the
// dynamic LDS (cap keys, a power of two) or, beyond that, in the part's global scratch.  min_part: parts up to this size are
// left to the rank kernel (the two kernels write disjoint positions of the dispatch list).
constexpr int kRankPartMax = 32768;
__global__ __launch_bounds__(1024) void sched_order_xcd_kernel(const unsigned long long* __restrict__ prefix,
                                                               const unsigned int* __restrict__ tile_cost,
                                                               const int2* __restrict__ items, int N, int chunks, int cap,
                                                               unsigned long long* __restrict__ g_keys, int g_stride,
                                                               int2* __restrict__ worklist, int min_part) {
    extern __shared__ unsigned long long s_keys_lds[];
    __shared__ int s_start[8 * 64 + 1];               // first item of every chunk (chunks <= 512)
    __shared__ int s_count[8];
    const int x = blockIdx.x;
    const double share = (double)prefix[N] / (double)chunks + 1e-9;
    for (int c = threadIdx.x; c <= chunks; c += blockDim.x) {
        // first i in [0, N] whose chunk is >= c (chunk numbers do not decrease with i)
        int lo = 0, hi = N;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sched_chunk_of(prefix, mid, share, chunks) < c) lo = mid + 1; else hi = mid;
        }
        s_start[c] = (c == chunks) ? N : lo;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        int n = 0;
        for (int c = threadIdx.x; c < chunks; c += 8) n += s_start[c + 1] - s_start[c];
        s_count[threadIdx.x] = n;
    }
    __syncthreads();
    const int n_x = s_count[x];
    if (n_x <= min_part) return;                      // (whole workgroup; the rank kernel places this part's items)
    int size = 1;
    while (size < n_x) size <<= 1;
    // a part that does not fit the LDS (one part can hold most of a group's cheap tiles when a few tiles carry most of
    // the cost) is sorted in global scratch instead: g_stride >= the next power of two of N keys per part; slow, rare
    unsigned long long* s_keys = size <= cap ? s_keys_lds : g_keys + (size_t)x * (size_t)g_stride;
    for (int i = threadIdx.x; i < size; i += blockDim.x) s_keys[i] = ~0ull;
    __syncthreads();
    int off = 0;
    for (int c = x; c < chunks; c += 8) {
        const int a = s_start[c], b = s_start[c + 1];
        for (int i = a + (int)threadIdx.x; i < b; i += blockDim.x) s_keys[off + (i - a)] = sched_key(tile_cost[i], i);
        off += b - a;
    }
    bitonic_sort_lds(s_keys, size);
    int m = s_count[0];
#pragma unroll
    for (int y = 1; y < 8; ++y) m = min(m, s_count[y]);
    for (int r = threadIdx.x; r < n_x; r += blockDim.x) {
        const int src = (int)(unsigned int)(s_keys[r] & 0xFFFFFFFFull);
        long long pos;
        if (r < m) {
            pos = 8LL * r + x;
        } else {
            pos = 8LL * m;
            for (int y = 0; y < 8; ++y) {
                pos += max(0, min(s_count[y], r) - m);                 // full rounds m .. r-1
                if (y < x && s_count[y] > r) pos += 1;                  // parts ahead of x in round r
            }
        }
        worklist[pos] = items[src];
    }
}

// minimum over the 64 lanes, the same value in every lane: four DPP butterfly steps inside the rows of 16 (always-valid
// sources), then the four row minima through scalar registers.  (The packing loop below runs this once per item: with
// __shfl_xor - six ds_bpermute round trips of ~100 cycles - it took 0.6 ms for 879 items, round 4.)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(v >> 32), CTRL, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    v = dpp_min_u64<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_min_u64<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_min_u64<0x141>(v);       // row_half_mirror
    v = dpp_min_u64<0x140>(v);       // row_mirror: every lane of a row holds the row's minimum
    auto row = [&](int l) {
        return ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l) << 32) |
               (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    };
    const unsigned long long a = row(0), b = row(16), c = row(32), d = row(48);
    const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

template <int CTRL>
__device__ __forceinline__ unsigned int dpp_min_u32(unsigned int v) {
    const unsigned int o = (unsigned int)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, 0xf, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
    v = dpp_min_u32<0xB1>(v);
    v = dpp_min_u32<0x4E>(v);
    v = dpp_min_u32<0x141>(v);
    v = dpp_min_u32<0x140>(v);
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// The packing loop of sched_order_pack_kernel, one wave: item k (longest first) to the least loaded bin with a free
// slot, lowest bin number on a tie.  Lane l owns bins l, 64 + l, ...; the two versions assign identically.
__device__ __forceinline__ void pack_bins_u64(const unsigned long long* s_keys, short* s_bin, short* s_tier, short* s_size,
                                              int N, int n_cu, int slots) {
    const int lane = threadIdx.x;
    unsigned long long load[8];
    int cnt[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) { load[b] = 0ull; cnt[b] = 0; }
    for (int k = 0; k < N; ++k) {
        const unsigned int cost = 0xFFFFFFFFu - (unsigned int)(s_keys[k] >> 32);
        unsigned long long best = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bin = b * 64 + lane;
            const unsigned long long cand = (bin < n_cu && cnt[b] < slots) ? ((load[b] << 10) | (unsigned long long)bin) : ~0ull;
            best = cand < best ? cand : best;
        }
        best = wave_min_u64(best);
        const int bin = (int)(best & 1023ull);
        if ((bin & 63) == lane) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b == (bin >> 6)) { s_tier[k] = (short)cnt[b]; load[b] += cost; cnt[b] += 1; }
            s_bin[k] = (short)bin;
        }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) if (b * 64 + lane < n_cu) s_size[b * 64 + lane] = (short)cnt[b];
}
// 32-bit keys (load << 12 | bin << 3 | items in the bin; all ones once the bin is full or absent): the wave minimum is
// then a scalar that names the bin, its lane, its tier and its load at once, so the winner's new key is scalar
// arithmetic and one select per register, and the (bin, tier) of item k0 + j is kept by lane j -
// no LDS and no divergent branch inside the loop.  While every load is zero, item k goes to bin k: the first n_cu
// items (of non-zero cost) are placed without the loop.
template <int NB>
__device__ __forceinline__ void pack_bins_u32(const unsigned long long* s_keys, short* s_bin, short* s_tier, short* s_size,
                                              int N, int n_cu, int slots) {
    const int lane = threadIdx.x;
    auto cost_of = [&](int k) { return k < N ? 0xFFFFFFFFu - (unsigned int)(s_keys[k] >> 32) : 0u; };
    const int first = N < n_cu ? N : n_cu;
    const bool seeded = cost_of(first - 1) > 0u;       // sorted longest-first: then all of the first tier are non-zero
    unsigned int key[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int bin = b * 64 + lane;
        key[b] = bin < n_cu ? (unsigned int)(bin << 3) : 0xFFFFFFFFu;
        if (seeded && bin < first) {
            key[b] = slots > 1 ? ((cost_of(bin) << 12) | (unsigned int)(bin << 3) | 1u) : 0xFFFFFFFFu;
            s_bin[bin] = (short)bin;
            s_tier[bin] = 0;
        }
    }
    for (int k0 = seeded ? first : 0; k0 < N; k0 += 64) {
        const int my_cost = (int)cost_of(k0 + lane);   // the costs of 64 items, one per lane
        const int n = N - k0 < 64 ? N - k0 : 64;
        int placed = 0;                                // lane j: (bin << 3 | tier) of item k0 + j
        for (int j = 0; j < n; ++j) {
            const unsigned int cost = (unsigned int)__builtin_amdgcn_readlane(my_cost, j);
            unsigned int best = key[0];
#pragma unroll
            for (int b = 1; b < NB; ++b) best = key[b] < best ? key[b] : best;
            best = wave_min_u32(best);                                 // uniform
            const int bin = (int)((best >> 3) & 511u), tier = (int)(best & 7u);
            const unsigned int next = tier + 1 < slots ? best + (cost << 12) + 1u : 0xFFFFFFFFu;
            const bool mine = (bin & 63) == lane;
#pragma unroll
            for (int b = 0; b < NB; ++b) key[b] = (mine && (bin >> 6) == b) ? next : key[b];
            placed = lane == j ? (int)(best & 4095u) : placed;
        }
        if (lane < n) { s_bin[k0 + lane] = (short)(placed >> 3); s_tier[k0 + lane] = (short)(placed & 7); }
    }
    // a bin's item count: what its key says, or `slots` once it is full
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b * 64 + lane < n_cu) s_size[b * 64 + lane] = (short)(key[b] == 0xFFFFFFFFu ? slots : (int)(key[b] & 7u));
}

// Launches of one round (every workgroup resident from the first cycle, nothing dispatched dynamically: the kernel
// lasts as long as the busiest CU): items sorted longest-first, each to the least loaded of the n_cu bins that still
// has a free slot, emitted bin-interleaved so that the dispatcher's round robin over the CUs rebuilds the bins.
// One workgroup of 256; N <= 1024 items, n_cu <= 512 bins (8 per lane of the packing wave).
__global__ __launch_bounds__(256) void sched_order_pack_kernel(const unsigned int* __restrict__ tile_cost,
                                                               const int2* __restrict__ items, int N, int n_cu,
                                                               int2* __restrict__ worklist) {
    __shared__ unsigned long long s_keys[1024];
    __shared__ short s_bin[1024], s_tier[1024];
    __shared__ short s_size[512];
    __shared__ int s_tier_off[8];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_keys[i] = i < N ? sched_key(tile_cost[i], i) : ~0ull;
    bitonic_sort_lds(s_keys, 1024);
    const int slots = (N + n_cu - 1) / n_cu;           // <= 4 (N <= 4 n_cu)
    // A bin's load stays below slots * (largest cost): when that fits 20 bits the packing loop runs on 32-bit keys,
    // ~40 instructions per item instead of ~160 (64-bit compares and selects, LDS stores under a divergent branch).
    const unsigned int max_cost = 0xFFFFFFFFu - (unsigned int)(s_keys[0] >> 32);
    const bool narrow = N > 0 && slots <= 7 && (unsigned long long)max_cost * (unsigned int)slots < (1ull << 20);
    if (threadIdx.x < 64) {
        if (narrow && n_cu <= 256) pack_bins_u32<4>(s_keys, s_bin, s_tier, s_size, N, n_cu, slots);
        else if (narrow) pack_bins_u32<8>(s_keys, s_bin, s_tier, s_size, N, n_cu, slots);
        else pack_bins_u64(s_keys, s_bin, s_tier, s_size, N, n_cu, slots);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int t = 0; t < slots && t < 8; ++t) {
            s_tier_off[t] = run;
            for (int b = 0; b < n_cu; ++b) run += s_size[b] > t ? 1 : 0;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const int bin = s_bin[k], t = s_tier[k];
        int rank = 0;
        for (int b = 0; b < bin; ++b) rank += s_size[b] > t ? 1 : 0;
        worklist[s_tier_off[t] + rank] = items[(int)(unsigned int)(s_keys[k] & 0xFFFFFFFFull)];
    }
}

// Single-round launches, packed per XCD (round 5).  Workgroup p of a launch runs on XCD p % 8: XCD x packs ITS tiles
// longest-first into its own n_cu / 8 CUs (one wave per XCD, one bin per lane: the loop of pack_bins_u32) and emits them
// at the positions k 8 + x; eight waves in parallel, ~35 us where the one-wave packing of 879 items over all CUs took
// 0.14 ms (0.43 before its 32-bit keys, 0.60 in round 4).  Which tiles are an XCD's:
//   local 1: a CONTIGUOUS run of the tile sequence worth an eighth of the cost (midpoint rule on the cost prefix; `chunks`
//     / 8 runs per XCD): its L2 then holds its own records only - 5.3 MB fetched per launch instead of 14.8 on a
//     per-list shard of 8 (879 workgroups), 5.5 instead of 21 on a one-list 500-900 cm^-1 cell (782).  But every wave of a
//     single round starts at the same time and walks its records in step with its neighbours, and with neighbours that
//     are neighbours in the SPECTRUM all CUs of an XCD ask for the same lines of the same L2 channels at once.  Measured
//     against the mixed order, same box: launches whose waves each own a span (4 spans per workgroup) 391 workgroups
//     +-0, 588 +-0, 879 -1 %; launches whose spans are shared by 2 or 4 waves (which read the same records again) 782
//     workgroups +10 %, 586 (merged shard of 16) +17 %, a per-list shard of 16 +9 % (the order inside the XCD - tiers
//     closed up, one position per CU and tier, bins by fill - changed nothing; 2, 4 or 8 runs per XCD neither).  So: the
//     launches with unsplit spans (the caller's rule; "accum_xcd_pack" 2 / 3 force either).
//   local 0: every 8th tile of the longest-first order - each XCD a like sample of the whole spectrum, which is what
//     the one packing over all CUs (sched_order_pack_kernel) gave it.
// A local launch falls back to the mixed order inside this kernel when a run does not fit the `m_cap` positions the
// host reserved per XCD (costs piled up in a few tiles) or when the busiest CU of some XCD carries more than `tol`
// percent above the mean of the eight by the cost model (an XCD whose run holds the spectrum's expensive tiles cannot
// hand any to another XCD's CUs).  Positions without a tile hold (0, -1): workgroups that exit at once.
__global__ __launch_bounds__(512) void sched_order_pack_xcd_kernel(const unsigned int* __restrict__ tile_cost,
                                                                   const int2* __restrict__ items, int N, int n_cu, int m_cap,
                                                                   int local, int chunks, int tol, int2* __restrict__ worklist) {
    __shared__ unsigned long long s_keys[1024];
    __shared__ unsigned long long s_pref[1024];          // inclusive prefix of the costs, in tile order
    __shared__ unsigned long long s_scan[8];
    __shared__ int s_end[9], s_cnt[8], s_retry;
    __shared__ unsigned int s_make[8];
    __shared__ short s_xcd[1024];
    __shared__ short s_place[1024];                      // k-th item of XCD x (slot x * 128 + k ... see at()) -> bin << 3 | tier
    __shared__ short s_inv[8][7 * 64];                   // position of the XCD -> its k-th item, -1 idle
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = wave, B = n_cu >> 3;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool mixed = local == 0 || attempt == 1;
        if (!mixed) {
            // cost prefix: two items per thread, a wave scan, the eight wave totals
            const unsigned int c0 = 2 * tid < N ? tile_cost[2 * tid] : 0u, c1 = 2 * tid + 1 < N ? tile_cost[2 * tid + 1] : 0u;
            unsigned long long run = (unsigned long long)c0 + c1;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned long long o = __shfl_up(run, d);
                if (lane >= d) run += o;
            }
            if (lane == 63) s_scan[wave] = run;
            if (tid < 8) s_cnt[tid] = 0;
            __syncthreads();
            unsigned long long base = 0ull, total = 0ull;
#pragma unroll
            for (int w = 0; w < 8; ++w) { base += w < wave ? s_scan[w] : 0ull; total += s_scan[w]; }
            s_pref[2 * tid + 1] = base + run;
            s_pref[2 * tid] = base + run - c1;
            __syncthreads();
            // XCD of item i: the midpoint of its cost falls into one of `chunks` equal shares of the total; share c is XCD c % 8's
            for (int i = tid; i < N; i += blockDim.x) {
                const unsigned long long cost = (unsigned long long)tile_cost[i];
                const unsigned long long mid = 2ull * (s_pref[i] - cost) + cost;        // 2 x midpoint
                unsigned long long c = total ? mid * (unsigned long long)chunks / (2ull * total) : 0ull;
                c = c >= (unsigned long long)chunks ? (unsigned long long)chunks - 1ull : c;
                s_xcd[i] = (short)(c & 7ull);
                atomicAdd(&s_cnt[(int)(c & 7ull)], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int run_n = 0, worst = 0;
                for (int y = 0; y < 8; ++y) { s_end[y] = run_n; run_n += s_cnt[y]; worst = max(worst, s_cnt[y]); }
                s_end[8] = run_n;
                s_retry = worst > m_cap ? 1 : 0;
            }
            __syncthreads();
            if (s_retry) continue;                       // (uniform: every thread reads the same flag)
        }
        // one sort: (by XCD,) longest first, ties by position
        __syncthreads();
        for (int i = tid; i < 1024; i += blockDim.x) {
            unsigned long long key = ~0ull;
            if (i < N)
                key = ((unsigned long long)(mixed ? 0 : s_xcd[i]) << 42) | ((unsigned long long)(0xFFFFFFFFu - tile_cost[i]) << 10) | (unsigned long long)i;
            s_keys[i] = key;
        }
        bitonic_sort_lds(s_keys, 1024);
        // wave x packs its items - the sorted positions [s_end[x], s_end[x + 1]), or x, x + 8, ... - into its bins = lanes 0 .. B - 1
        const int first = mixed ? x : s_end[x], stride = mixed ? 8 : 1;
        const int n_x = mixed ? (N - x + 7) / 8 : s_end[x + 1] - s_end[x];
        auto at = [&](int k) { return first + k * stride; };
        const int slots = (n_x + B - 1) / B;             // <= 7 (sched_xcd_positions)
        auto cost_at = [&](int k) { return k < n_x ? 0xFFFFFFFFu - (unsigned int)((s_keys[at(k)] >> 10) & 0xFFFFFFFFull) : 0u; };
        // loads as 23-bit numbers: costs scaled down where slots x (largest cost) needs more (the order of nearly equal
        // loads is all that changes)
        int shift = 0;
        while (n_x > 0 && (((unsigned long long)cost_at(0) * (unsigned int)(slots > 0 ? slots : 1)) >> shift) >= (1ull << 23)) ++shift;
        const int seed_n = n_x < B ? n_x : B;
        const bool seeded = n_x > 0 && (cost_at(seed_n - 1) >> shift) > 0u;     // while every load is zero, item k goes to bin k
        unsigned int key = lane < B ? (unsigned int)(lane << 3) : 0xFFFFFFFFu;  // load << 9 | bin << 3 | items in the bin
        if (seeded && lane < seed_n) {
            key = slots > 1 ? (((cost_at(lane) >> shift) << 9) | (unsigned int)(lane << 3) | 1u) : 0xFFFFFFFFu;
            s_place[at(lane)] = (short)(lane << 3);
        }
        int full_cnt = (seeded && lane < seed_n && slots == 1) ? 1 : 0;         // what an all-ones key no longer says
        unsigned int my_load = (seeded && lane < seed_n) ? cost_at(lane) : 0u;  // the bin's load, unscaled
        for (int k0 = seeded ? seed_n : 0; k0 < n_x; k0 += 64) {
            const int my_raw = (int)cost_at(k0 + lane);
            const int my_cost = (int)((unsigned int)my_raw >> shift);
            const int n = n_x - k0 < 64 ? n_x - k0 : 64;
            int placed = 0;
            for (int j = 0; j < n; ++j) {
                const unsigned int cost = (unsigned int)__builtin_amdgcn_readlane(my_cost, j);
                const unsigned int raw = (unsigned int)__builtin_amdgcn_readlane(my_raw, j);
                const unsigned int best = wave_min_u32(key);
                const int bin = (int)((best >> 3) & 63u), tier = (int)(best & 7u);
                const bool last = tier + 1 >= slots;
                const unsigned int next = last ? 0xFFFFFFFFu : best + (cost << 9) + 1u;
                if (lane == bin) { key = next; full_cnt = last ? tier + 1 : 0; my_load += raw; }
                placed = lane == j ? (int)(best & 511u) : placed;
            }
            if (lane < n) s_place[at(k0 + lane)] = (short)placed;
        }
        if (!mixed) {
            unsigned int mk = my_load;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) mk = max(mk, (unsigned int)__shfl_xor((int)mk, d));
            if (lane == 0) s_make[x] = mk;
            __syncthreads();
            unsigned long long sum = 0ull;
            unsigned int worst = 0u;
#pragma unroll
            for (int y = 0; y < 8; ++y) { sum += s_make[y]; worst = max(worst, s_make[y]); }
            if (tol >= 0 && (unsigned long long)worst * 800ull > sum * (unsigned long long)(100 + tol)) continue;   // (uniform)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int my_size = lane < B ? (key == 0xFFFFFFFFu ? full_cnt : (int)(key & 7u)) : 0;
        // tier by tier, closed up, the bins in bin order
        for (int k = lane; k < m_cap; k += 64) s_inv[x][k] = (short)-1;
        unsigned long long tier_mask[8];
        int tier_off[8];
        int off = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            tier_mask[t] = __ballot(my_size > t);
            tier_off[t] = off;
            off += __popcll(tier_mask[t]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n_x; k += 64) {
            const int pl = s_place[at(k)], bin = pl >> 3, t = pl & 7;
            unsigned long long m = 0ull;
            int o = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) { m = u == t ? tier_mask[u] : m; o = u == t ? tier_off[u] : o; }
            s_inv[x][o + __popcll(m & ((1ull << bin) - 1ull))] = (short)k;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < m_cap; k += 64) {
            const int src = s_inv[x][k];
            worklist[(size_t)k * 8 + x] = src >= 0 ? items[(int)(s_keys[at(src)] & 1023ull)] : make_int2(0, -1);
        }
        return;
    }
}

// (sched_xcd_positions, sched_launch_items, sched_device_supported, sched_key_stride, sched_scratch_bytes: lbl_launch_shapes.h -
// pure host arithmetic shared with the sanitizer harness of the host shim)
void launch_schedule_build(const SchedJob* d_jobs, int n_jobs, int total_spans, int total_tiles, int R, int spans_per_tile,
                           long long far_reach, double cost_near, double cost_edge, double cost_far, double cost_fixed,
                           int n_cu, int32_t* tabs, void* scratch, int2* worklist, hipStream_t s, int xcd_chunks, bool xcd_pack,
                           int single_round_chunks, int xcd_tol, int xcd_local) {
    if (total_spans <= 0 || total_tiles <= 0) return;
    const size_t n = (size_t)total_tiles;
    char* base = (char*)scratch;
    unsigned int* tile_cost = (unsigned int*)base;                      base += (n * 4 + 255) & ~(size_t)255;
    int2* items = (int2*)base;                                          base += (n * 8 + 255) & ~(size_t)255;
    unsigned long long* prefix = (unsigned long long*)base;             base += ((n + 1) * 8 + 255) & ~(size_t)255;
    unsigned long long* g_keys = (unsigned long long*)base;
    (void)hipMemsetAsync(tile_cost, 0, (size_t)total_tiles * sizeof(unsigned int), s);
    hipLaunchKernelGGL(sched_spans_kernel, dim3((total_spans + 255) / 256), dim3(256), 0, s, d_jobs, n_jobs, total_spans, R,
                       spans_per_tile, far_reach, cost_near, cost_edge, cost_far, cost_fixed, tabs, tile_cost, items);
    if (total_tiles <= 4 * n_cu) {
        const int m_cap = xcd_pack ? sched_xcd_positions(total_tiles, n_cu) : 0;
        if (m_cap > 0) {
            // (runs per XCD: 1 by default.  A tile's records reach ~20 tiles to either side, so only runs much longer than that
            // keep an L2 to its own records: one run of ~50-110 tiles fetched 5.3-5.5 MB where 2 / 4 / 8 runs fetched 6.0 / 7.1 /
            // 9.2 and the packing over all CUs 14.8-21; the step times of 1, 2, 4 and 8 runs did not differ on the shards.)
            const int local = xcd_local == 1 || (xcd_local < 0 && spans_per_tile >= 4) ? 1 : 0;         // (auto: unsplit spans only)
            hipLaunchKernelGGL(sched_order_pack_xcd_kernel, dim3(1), dim3(512), 0, s, tile_cost, items, total_tiles, n_cu, m_cap,
                               local, 8 * single_round_chunks, xcd_tol, worklist);
        } else {
            hipLaunchKernelGGL(sched_order_pack_kernel, dim3(1), dim3(256), 0, s, tile_cost, items, total_tiles, n_cu, worklist);
        }
        return;
    }
    hipLaunchKernelGGL(scan_counts_kernel, dim3(scan_blocks(total_tiles)), dim3(1024), 0, s, tile_cost, total_tiles, prefix);
    // the parts' keys to global scratch, then one thread per key ranks it inside its part (see sched_rank_xcd_kernel)
    const int chunks = 8 * (xcd_chunks > 0 && xcd_chunks <= 64 ? xcd_chunks : 32);
    const int stride = sched_key_stride(total_tiles);
    int* part_count = reinterpret_cast<int*>(g_keys + (size_t)8 * (size_t)stride);         // (the scratch block's last 256 bytes)
    hipLaunchKernelGGL(sched_parts_kernel, dim3(8), dim3(1024), 0, s, prefix, tile_cost, total_tiles, chunks, g_keys, stride, part_count);
    const int part_tiles = (std::min(total_tiles, kRankPartMax) + 1023) / 1024;
    hipLaunchKernelGGL(sched_tile_sort_kernel, dim3(part_tiles, 8), dim3(256), 0, s, g_keys, stride, part_count, kRankPartMax);
    hipLaunchKernelGGL(sched_rank_xcd_kernel, dim3(part_tiles, 8), dim3(1024), 0, s, g_keys, stride, part_count, items, worklist, kRankPartMax);
    if (total_tiles > kRankPartMax) {
        // (a part of that size sorts in its own slice of the key scratch, which the rank kernel does not read for such a part)
        int cap = 16384;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(sched_order_xcd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                cap * (int)sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            cap = 4096;
        }
        hipLaunchKernelGGL(sched_order_xcd_kernel, dim3(8), dim3(1024), cap * sizeof(unsigned long long), s, prefix, tile_cost, items,
                           total_tiles, chunks, cap, g_keys, stride, worklist, kRankPartMax);
    }
}

// ----------------------------------------------------------------------------------------
// K3: np.interp from linspace(min,max,n_work) onto linspace(min,max,n_base)
//     (pyradClasses.py:401-405, 159-162)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void regrid_kernel(const double* __restrict__ work, long long n_work,
                                                     double* __restrict__ out, long long n_base,
                                                     double start, double stop, double step_w, double step_b) {
#pragma clang fp contract(off)
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_base) return;
    const double x = linspace_at(j, n_base, start, stop, step_b);
    double res;
    if (n_work == 1) {
        // np.interp with a single sample: x > xp[0] -> right, x < xp[0] -> left, equal -> fp[0]
        res = work[0];
    } else if (x > stop) {
        res = work[n_work - 1];
    } else if (x < start) {
        res = work[0];
    } else {
        long long i = (long long)((x - start) / step_w);
        if (i < 0) i = 0;
        if (i > n_work - 1) i = n_work - 1;
        while (i > 0 && linspace_at(i, n_work, start, stop, step_w) > x) --i;
        while (i < n_work - 1 && linspace_at(i + 1, n_work, start, stop, step_w) <= x) ++i;
        const double xi = linspace_at(i, n_work, start, stop, step_w);
        if (i == n_work - 1 || xi == x) {
            res = work[i];
        } else {
            const double xi1 = linspace_at(i + 1, n_work, start, stop, step_w);
            const double yi = work[i], yi1 = work[i + 1];
            const double slope = (yi1 - yi) / (xi1 - xi);
            res = slope * (x - xi) + yi;
            if (isnan(res)) {
                res = slope * (x - xi1) + yi1;
                if (isnan(res) && yi == yi1) res = yi;
            }
        }
    }
    out[j] = res;
}

// ----------------------------------------------------------------------------------------
// K4: fused layer sweep
// ----------------------------------------------------------------------------------------
// Cross sections are read in batches of NB independent loads (the term list is wave-uniform: pointers and
// flags come from the argument block by scalar loads; explicit global address space, so that a load does not
// wait for the scalar ones).  The first version walked molecules and isotopologues with dependent scalar
// loads and waited for every value before asking for the next: 0.52 of the HBM rate.
typedef const double __attribute__((address_space(1)))* GlobalF64;
__device__ __forceinline__ double load_global_f64(const double* p, long long j) {
    return ((GlobalF64)(unsigned long long)p)[j];
}

template <bool NT, int NP, bool BUDGET = false>
__global__ __launch_bounds__(256) void layer_sweep_kernel(const SweepArgs A) {
#pragma clang fp contract(off)
    // NP grid points per thread (2: 16-byte accesses; the host gives this instantiation an even first point and count)
    constexpr int NB = 4;
    typedef double vec __attribute__((ext_vector_type(NP)));
    auto ld = [&](const double* p, long long j) {
        vec v;
        if (NP == 1) v[0] = NT ? __builtin_nontemporal_load(p + j) : load_global_f64(p, j);
        else v = NT ? __builtin_nontemporal_load(reinterpret_cast<const vec*>(p + j)) : *reinterpret_cast<const vec*>(p + j);
        return v;
    };
    auto st = [&](double* p, long long j, vec v) {
        if (NP == 1) { if (NT) __builtin_nontemporal_store(v[0], p + j); else p[j] = v[0]; }
        else { if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vec*>(p + j)); else *reinterpret_cast<vec*>(p + j) = v; }
    };
    const long long stride = (long long)gridDim.x * blockDim.x * NP;
    const long long jend = A.first + A.count;
    const int n_full = A.n_iso - A.n_iso % NB;
    for (long long j = A.first + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * NP; j < jend; j += stride) {
        // Layer.absCoef (pyradClasses.py:707-712): zeros + sum over molecules of
        // Molecule.absCoef = crossSection * concentration * P / 1E4 / k / T (pyradClasses.py:583),
        // Molecule.crossSection = zeros + sum over isotopologues (pyradClasses.py:566-571)
        vec kk = (vec)(0.0), xs = (vec)(0.0);
        auto term = [&](int t, vec v) {
            xs += v;
            if (A.term_flags[t] & TERM_LAST_MOL) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    kk[p] += BUDGET ? xs[p] * A.term_factor[t] : abs_coef_term(xs[p], A.term_conc[t], A.P, A.T, A.rT);
                    xs[p] = 0.0;
                }
            }
        };
        for (int t0 = 0; t0 < n_full; t0 += NB) {
            vec v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = ld(A.xsec[t0 + u], j);
#pragma unroll
            for (int u = 0; u < NB; ++u) term(t0 + u, v[u]);
        }
        {
            vec v[NB];
#pragma unroll
            for (int u = 0; u < NB - 1; ++u) v[u] = n_full + u < A.n_iso ? ld(A.xsec[n_full + u], j) : (vec)(0.0);
#pragma unroll
            for (int u = 0; u < NB - 1; ++u) if (n_full + u < A.n_iso) term(n_full + u, v[u]);
        }
        if (A.abs_coef) st(A.abs_coef, j, kk);
        vec tr;
#pragma unroll
        for (int p = 0; p < NP; ++p) tr[p] = BUDGET ? exp_neg_budget(kk[p] * A.depth) : exp(-kk[p] * A.depth);     // pyradClasses.py:716
        if (A.trans) st(A.trans, j, tr);
        if (A.I_out) {
            vec out;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const double nu = linspace_at(j + p, A.n, A.start, A.stop, A.step);
                double B, Iin;
                if (BUDGET) {
                    B = planck_budget(nu, A.pa, A.pbkT);
                    Iin = A.I_in ? A.I_in[j + p] : planck_budget(nu, A.pa, A.pbk_surface);
                } else {
                    double pa_n, pb_n;
                    planck_point(nu, A.pa, A.pb, pa_n, pb_n);
                    B = planck_at(pa_n, pb_n, A.T, A.rT);                       // Layer.planck(self.T)
                    Iin = A.I_in ? A.I_in[j + p] : planck_at(pa_n, pb_n, A.surface_T, A.r_surface_T);
                }
                const double transmitted = tr[p] * Iin;                     // pyradClasses.py:785
                const double emitted = (1.0 - tr[p]) * B;                   // pyradClasses.py:786
                out[p] = transmitted + emitted;
            }
            st(A.I_out, j, out);
        }
    }
}

// ----------------------------------------------------------------------------------------
// K5: column fold (pyradClasses.py:784-787 applied layer after layer)
// ----------------------------------------------------------------------------------------
template <bool BUDGET>
__global__ __launch_bounds__(256) void column_sweep_kernel(const ColumnArgs* __restrict__ Ap) {
#pragma clang fp contract(off)
    const ColumnArgs& A = *Ap;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long jend = A.first + A.count;
    for (long long j = A.first + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < jend; j += stride) {
        const double nu = linspace_at(j, A.n, A.start, A.stop, A.step);
        double I = A.I_in ? A.I_in[j] : (BUDGET ? planck_budget(nu, A.pa, A.pbk_surface)
                                                : planck_wn(nu, A.surface_T, A.r_surface_T, A.pa, A.pb));
        for (int l = 0; l < A.n_layers; ++l) {
            const double tr = A.trans[l][j];
            const double B = BUDGET ? planck_budget(nu, A.pa, A.pbkT[l]) : planck_wn(nu, A.layer_T[l], A.r_layer_T[l], A.pa, A.pb);
            const double transmitted = tr * I;
            const double emitted = (1.0 - tr) * B;
            I = transmitted + emitted;
        }
        A.I_out[j] = I;
    }
}

// K5b: column step straight from the cross sections.  For every grid point the layers are visited
// bottom to top: absorption coefficient and transmittance exactly as layer_sweep_kernel computes
// them, then the fold of column_sweep_kernel.  One pass over the cross sections replaces one sweep
// launch per layer plus the fold, and the per-layer arrays are written only if asked for.
// The column is a flat list of terms (one per cross-section array, bottom layer first; flags mark the last
// term of a molecule and of a layer).  NB terms are loaded at once, independent of one another; everything
// that depends on the grid point only (2E8 h c^2 n^3 and 100 h c n / k of pyradPlanck.py:41-42) is computed
// once per point, not once per layer.
// exp(x) for 0 <= x <= 1e-3 (the Planck exponent's growth over the 1-3 grid steps between a thread's points): degree-4
// Taylor polynomial, remainder x^5 / 120 <= 8.4e-18 relative
__device__ __forceinline__ double expm1_tiny(double x) {
    return x * fma(x, fma(x, fma(x, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0);
}

// KFOLD (round 6): the fold over absorption coefficients (lbl_column_fold_dev, the column handle) - every term IS a layer, its
// factor 1: the per-molecule sums are skipped (0 + v = v, v * 1 = v: the same bits) and the flags are never read.
template <int NP, bool BUDGET = false, bool KFOLD = false>
__global__ __launch_bounds__(256, (NP == 4 ? 3 : 1)) void column_step_kernel(const ColumnStepArgs* __restrict__ Ap, long long first, long long count) {
#pragma clang fp contract(off)
    // NP grid points per thread (2: 16-byte loads, 4: 32-byte; the host gives these instantiations a first point and a count
    // that are multiples of NP)
    // terms per batch of loads (two batches in flight).  Four points per thread: 2 - the fold over absorption coefficients 128 VGPRs,
    // four waves per SIMD (135 us for the 30-layer column), the step from cross sections 138 / three
    constexpr int NB = NP == 4 ? 2 : 6;     // (the fold over absorption coefficients with 3: 158 VGPRs, three waves per SIMD, no faster; 4 spills)
    typedef double vec __attribute__((ext_vector_type(NP)));
    typedef const vec __attribute__((address_space(1)))* GlobalVec;
    const ColumnStepArgs& A = *Ap;
    const long long stride = (long long)gridDim.x * blockDim.x * NP;
    const long long jend = first + count;
    const int n_terms = A.n_terms;
    const int n_full = n_terms - n_terms % NB;
    const bool layer_arrays = A.layer_arrays != 0;          // (streaming / non-temporal loads measured 18 % SLOWER here: 536 vs 455 us)
    auto load = [&](const double* p, long long j) {
        vec v;
        if (NP == 1) v[0] = load_global_f64(p, j);
        else v = *(GlobalVec)(unsigned long long)(p + j);
        return v;
    };
    for (long long j = first + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * NP; j < jend; j += stride) {
        double pa_n[NP], pb_n[NP], I[NP], kk[NP], xs[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double nu = linspace_at(j + p, A.n, A.start, A.stop, A.step);
            if (BUDGET) {
                pa_n[p] = A.pa * (nu * nu * nu); pb_n[p] = nu;           // (budget: the exponent is nu * term_pbkT)
                I[p] = A.I_in ? A.I_in[j + p] : planck_budget(nu, A.pa, A.pbk_surface);
            } else {
                planck_point(nu, A.pa, A.pb, pa_n[p], pb_n[p]);
                I[p] = A.I_in ? A.I_in[j + p] : planck_at(pa_n[p], pb_n[p], A.surface_T, A.r_surface_T);
            }
            kk[p] = 0.0; xs[p] = 0.0;
        }
        // Round 6, default arithmetic with several points per thread: ONE exp per thread and layer for the Planck function.
        // exp(nu_p c) = exp(nu_0 c) exp((nu_p - nu_0) c): the difference of two neighbouring grid wavenumbers is exact in
        // fp64 and (nu_p - nu_0) c <= 1e-3 for every layer (checked here with the column's largest c = 100 h c / k / T_min),
        // so the second factor is a degree-4 polynomial (7 instructions instead of the 17 of exp; 1-2 ulp).  Also only
        // where no lane of the wave can meet a special case - exponent above 700, exp(b) - 1 not positive (nu = 0), NaN
        // wavenumbers - which keep the general expression with its selects and its IEEE division (wave-uniform branch).
        const double dnu_last = BUDGET ? pb_n[NP - 1] - pb_n[0] : 0.0;
        const bool plain = BUDGET && NP > 1 && pb_n[0] * A.pbkT_min >= 1e-6 && pb_n[NP - 1] * A.pbkT_max <= 690.0
                           && dnu_last * A.pbkT_max <= 1e-3 && dnu_last >= 0.0;
        const bool fast = BUDGET && NP > 1 && __builtin_amdgcn_ballot_w64(!plain) == 0ull;
        int l = 0;
        // (arr_tag: std::true_type - per-layer arrays may be asked for, looked up per term; false_type - known to be absent: the
        //  wave-uniform test per point otherwise splits the four points' exponentials into separate basic blocks, round 6)
        auto term = [&](auto fast_tag, auto arr_tag, int t, vec v) {
            constexpr bool FAST = decltype(fast_tag)::value;
            constexpr bool ARRAYS = decltype(arr_tag)::value;
            const int f = KFOLD ? (TERM_LAST_MOL | TERM_LAST_LAYER) : A.term_flags[t];
            if (!KFOLD) {
#pragma unroll
                for (int p = 0; p < NP; ++p) xs[p] += v[p];
            }
            if (LBL_ABLATE(A, 16)) { I[0] += v[0]; return; }       // (diagnostic builds: memory traffic only)
            if (!KFOLD && (f & TERM_LAST_MOL)) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    kk[p] += BUDGET ? xs[p] * A.term_factor[t] : abs_coef_term(xs[p], A.term_conc[t], A.term_P[t], A.term_T[t], A.term_rT[t]);
                    xs[p] = 0.0;
                }
            }
            if (f & TERM_LAST_LAYER) {
                double E0 = 0.0;
                if (FAST) E0 = exp_clamped(pb_n[0] * A.term_pbkT[t]);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const double kp = KFOLD ? v[p] : kk[p];
                    const double tr = BUDGET ? exp_neg_budget(kp * A.term_depth[t]) : exp(-kp * A.term_depth[t]);
                    if (ARRAYS && layer_arrays) {
                        if (A.abs_coef[l]) A.abs_coef[l][j + p] = kp;
                        if (A.trans[l]) A.trans[l][j + p] = tr;
                    }
                    double B;
                    if (FAST) {
                        const double E = p == 0 ? E0 : fma(E0, expm1_tiny((pb_n[p] - pb_n[0]) * A.term_pbkT[t]), E0);
                        B = pa_n[p] * rcp_newton(E - 1.0);
                    } else if (BUDGET) {
                        const double b = pb_n[p] * A.term_pbkT[t];
                        const double e = exp_clamped(fmin(b, 700.0)) - 1.0;
                        B = (b > 700.0) ? 0.0 : (e > 0.0 ? pa_n[p] * rcp_newton(fmax(e, 1e-300)) : pa_n[p] / e);
                        B = b != b ? b : B;
                    } else {
                        B = planck_at(pa_n[p], pb_n[p], A.term_T[t], A.term_rT[t]);
                    }
                    if (FAST) {
                        // tr I + (1 - tr) B with the last product and sum as one fma (three instructions instead of four).
                        // (B + tr (I - B) would be two, but loses I against B: with I < 1e-16 B behind a layer of optical
                        // depth ~1e-16 the result tr I + (1 - tr) B ~ I + 1e-16 B would come out as 1e-16 B alone.)
                        I[p] = fma(tr, I[p], (1.0 - tr) * B);
                    } else {
                        const double transmitted = tr * I[p];
                        const double emitted = (1.0 - tr) * B;
                        I[p] = transmitted + emitted;
                    }
                    kk[p] = 0.0;
                }
                if (ARRAYS && layer_arrays) ++l;
            }
        };
        // two batches of NB loads in flight: the next batch is requested before the current one is consumed
        vec cur[NB], nxt[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) { cur[u] = (vec)(0.0); nxt[u] = (vec)(0.0); }
        if (n_full > 0) {
#pragma unroll
            for (int u = 0; u < NB; ++u) cur[u] = LBL_ABLATE(A, 32) ? (vec)(1e-22 * (double)(j & 7)) : load(A.xsec[u], j);
        }
        auto walk = [&](auto fast_tag, auto arr_tag) {
            for (int t0 = 0; t0 < n_full; t0 += NB) {
                if (t0 + NB < n_full) {
#pragma unroll
                    for (int u = 0; u < NB; ++u) nxt[u] = LBL_ABLATE(A, 32) ? (vec)(1e-22 * (double)(j & 7)) : load(A.xsec[t0 + NB + u], j);
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) term(fast_tag, arr_tag, t0 + u, cur[u]);
#pragma unroll
                for (int u = 0; u < NB; ++u) cur[u] = nxt[u];
            }
            for (int t = n_full; t < n_terms; ++t) term(fast_tag, arr_tag, t, load(A.xsec[t], j));
        };
        if (BUDGET && NP > 1 && fast && !layer_arrays) walk(std::true_type{}, std::false_type{});
        else if (BUDGET && NP > 1 && fast) walk(std::true_type{}, std::true_type{});
        else walk(std::false_type{}, std::true_type{});
#pragma unroll
        for (int p = 0; p < NP; ++p) A.I_out[j + p] = I[p];
    }
}

__global__ __launch_bounds__(256) void planck_kernel(double* __restrict__ out, long long n, double start,
                                                     double stop, double step, double T, double rT, double pa, double pb) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = planck_wn(linspace_at(j, n, start, stop, step), T, rT, pa, pb);
}

// ----------------------------------------------------------------------------------------
// K6: band integral, sum(nan_to_num(y)) (pyradClasses.py:26-29); fixed reduction tree
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ double nan_to_num(double v) {
    if (isnan(v)) return 0.0;
    if (isinf(v)) return v > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// every block sums a fixed, contiguous slice in a fixed order -> deterministic partials
__global__ __launch_bounds__(256) void band_partial_kernel(const double* __restrict__ y, long long n,
                                                           double* __restrict__ partial, long long per_block) {
    __shared__ double sh[4];
    const long long lo = (long long)blockIdx.x * per_block;
    const long long hi = min(lo + per_block, n);
    double s = 0.0;
    for (long long j = lo + threadIdx.x; j < hi; j += blockDim.x) s += nan_to_num(y[j]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void band_final_kernel(const double* __restrict__ partial, int n_partial,
                                                         double* __restrict__ result) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int j = threadIdx.x; j < n_partial; j += blockDim.x) s += partial[j];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) result[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ----------------------------------------------------------------------------------------
// K7: line survey (pyradClasses.py:409-428): S added into the bin of each line, in line order
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void line_survey_kernel(const double* __restrict__ nu, const double* __restrict__ sw,
                                                          int n_lines, double range_min, double resolution,
                                                          double* __restrict__ out, long long n_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const long long c = (long long)((nu[i] - range_min) / resolution);
    if (c < 0 || c > n_base - 1) return;
    if (i > 0 && (long long)((nu[i - 1] - range_min) / resolution) == c) return;   // not the first line of its bin
    double s = 0.0;                           // lineSurvey starts from zeros (pyradClasses.py:416)
    for (int k = i; k < n_lines; ++k) {
        if ((long long)((nu[k] - range_min) / resolution) != c) break;
        s = s + sw[k];
    }
    out[c] = s;
}

// ----------------------------------------------------------------------------------------
// launchers (called from lbl_api.hip)
// ----------------------------------------------------------------------------------------
void launch_line_prep(const PrepJob* d_jobs, int n_jobs, int max_lines, hipStream_t s) {
    if (n_jobs <= 0 || max_lines <= 0) return;
    dim3 grid((max_lines + 255) / 256, n_jobs);
    hipLaunchKernelGGL(line_prep_kernel, grid, dim3(256), 0, s, d_jobs);
}

void launch_line_prep_merged(const PrepJob* d_lists, const MergedPrep* d_jobs, int n_jobs, int max_total, hipStream_t s) {
    if (n_jobs <= 0 || max_total <= 0) return;
    dim3 grid((max_total + 255) / 256, n_jobs);
    hipLaunchKernelGGL(line_prep_merged_kernel, grid, dim3(256), 0, s, d_lists, d_jobs);
}

void launch_line_quantities(const PrepJob* d_job, int n_lines, long long* index, double* lhw, double* ghw, double* intensity,
                            int32_t* regime, hipStream_t s) {
    if (n_lines <= 0) return;
    hipLaunchKernelGGL(line_quantities_kernel, dim3((n_lines + 255) / 256), dim3(256), 0, s, d_job, index, lhw, ghw, intensity, regime);
}

template <int R>
static void launch_accum_scalar(const AccumJob* d_jobs, int n_jobs, int max_tiles, int variant, hipStream_t s) {
    dim3 grid(((max_tiles + 7) / 8) * 8, n_jobs);
    switch (variant) {
#ifdef LBL_DIAG      // (1: running fraction, 2: + Gaussian recurrence, both through the scalar cache: superseded comparison kernels)
        case 1: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 2, 0>), grid, dim3(256), 0, s, d_jobs); break;
        case 2: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 2, 1>), grid, dim3(256), 0, s, d_jobs); break;
#endif
        default: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 0, 0>), grid, dim3(256), 0, s, d_jobs); break;   // 0: IEEE divide + exp per pair
    }
}

template <int R, int NT>
static void launch_accum_lds(const AccumJob* d_jobs, int n_jobs, int max_tiles, int LS, const int2* worklist,
                             int total_tiles, hipStream_t s, int gauss_run = 16) {
    dim3 grid(((max_tiles + 7) / 8) * 8, n_jobs);
    if (worklist) {
        if (total_tiles <= 0) return;
        grid = dim3(total_tiles, 1);
    }
#ifdef LBL_DIAG        // (scripts/cost_fit.py): unused dynamic LDS to limit the workgroups resident per CU
    static const int pad = getenv("LBL_DIAG_LDS_PAD") ? atoi(getenv("LBL_DIAG_LDS_PAD")) : 0;
#else
    constexpr int pad = 0;
#endif
    switch (LS) {
        case 8: hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 8, NT>), grid, dim3(512), pad, s, d_jobs, worklist); break;
        case 4: hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 4, NT>), grid, dim3(256), pad, s, d_jobs, worklist); break;
        case 2: hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 2, NT>), grid, dim3(256), pad, s, d_jobs, worklist); break;
        default:
            if constexpr (R == 4 && NT > 0) {
                if (gauss_run == 32) {        // the three-waves-per-SIMD build; exact mode: its series starts at FF_MID half-spans (38 terms)
                    hipLaunchKernelGGL((xsec_accumulate_lds_kernel<4, 1, (NT == FF_NT ? FF_NT_MID : NT), 32>), grid, dim3(256), pad, s, d_jobs, worklist);
                    break;
                }
            }
            hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 1, NT>), grid, dim3(256), pad, s, d_jobs, worklist); break;
    }
}

void launch_accumulate_skew(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, const int2* worklist, int total_tiles,
                            hipStream_t s, int LS) {
    if (n_jobs <= 0 || max_tiles <= 0) return;
    dim3 grid(((max_tiles + 7) / 8) * 8, n_jobs);
    if (worklist) {
        if (total_tiles <= 0) return;
        grid = dim3(total_tiles, 1);
    }
    if (R == 8 && LS == 4) { hipLaunchKernelGGL((xsec_accumulate_skew_kernel<8, 4>), grid, dim3(256), 0, s, d_jobs, worklist); return; }
    if (R == 8 && LS == 2) { hipLaunchKernelGGL((xsec_accumulate_skew_kernel<8, 2>), grid, dim3(256), 0, s, d_jobs, worklist); return; }
    switch (R) {
        case 8: hipLaunchKernelGGL((xsec_accumulate_skew_kernel<8>), grid, dim3(256), 0, s, d_jobs, worklist); break;
        case 2: hipLaunchKernelGGL((xsec_accumulate_skew_kernel<2>), grid, dim3(256), 0, s, d_jobs, worklist); break;
        case 1: hipLaunchKernelGGL((xsec_accumulate_skew_kernel<1>), grid, dim3(256), 0, s, d_jobs, worklist); break;
        default: hipLaunchKernelGGL((xsec_accumulate_skew_kernel<4>), grid, dim3(256), 0, s, d_jobs, worklist); break;
    }
}

// far-field threshold (in half-spans of 32 R points) and the cost of a far line relative to a near
// one (3 instructions per series term and 64 lines against 5 R per line), for the host's schedule
void accumulate_far_field_params(int R, int* far_half_spans, double* far_cost, int budget) {
    *far_half_spans = budget ? FF_FAR_BUDGET : FF_FAR;
    *far_cost = (3.0 * 17.0 + 12.0) / 64.0 / (5.0 * R);           // (17 terms: the average over a +-39 half-span window, either mode)
}


void launch_accumulate(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, int LS, int variant,
                       const int2* worklist, int total_tiles, hipStream_t s, int budget, int gauss_run) {
    if (n_jobs <= 0 || max_tiles <= 0) return;
    if (variant >= 5 && budget) {
        switch (R) {
            case 1: launch_accum_lds<1, FF_NT_BUDGET>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 2: launch_accum_lds<2, FF_NT_BUDGET>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 4: launch_accum_lds<4, FF_NT_BUDGET>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s, gauss_run); break;
            default: launch_accum_lds<8, FF_NT_BUDGET>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
        }
        return;
    }
    if (variant >= 5) {
        switch (R) {
            case 1: launch_accum_lds<1, FF_NT>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 2: launch_accum_lds<2, FF_NT>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 4: launch_accum_lds<4, FF_NT>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s, gauss_run); break;
            default: launch_accum_lds<8, FF_NT>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
        }
        return;
    }
    if (variant >= 3) {
        switch (R) {
            case 1: launch_accum_lds<1, 0>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 2: launch_accum_lds<2, 0>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            case 4: launch_accum_lds<4, 0>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
            default: launch_accum_lds<8, 0>(d_jobs, n_jobs, max_tiles, LS, worklist, total_tiles, s); break;
        }
        return;
    }
    switch (R) {
        case 1: launch_accum_scalar<1>(d_jobs, n_jobs, max_tiles, variant, s); break;
        case 2: launch_accum_scalar<2>(d_jobs, n_jobs, max_tiles, variant, s); break;
        case 4: launch_accum_scalar<4>(d_jobs, n_jobs, max_tiles, variant, s); break;
        default: launch_accum_scalar<8>(d_jobs, n_jobs, max_tiles, variant, s); break;
    }
}

void launch_regrid(const double* work, long long n_work, double* out, long long n_base, double start,
                   double stop, hipStream_t s) {
    if (n_base <= 0) return;
    const double step_w = n_work > 1 ? (stop - start) / (double)(n_work - 1) : 0.0;
    const double step_b = n_base > 1 ? (stop - start) / (double)(n_base - 1) : 0.0;
    hipLaunchKernelGGL(regrid_kernel, dim3((unsigned)((n_base + 255) / 256)), dim3(256), 0, s, work, n_work, out,
                       n_base, start, stop, step_w, step_b);
}

static int sweep_blocks(long long n) {       // (one point per thread on a larger grid measured no faster: 442 vs 433 us for the column)
#ifdef LBL_DIAG
    static const long long cap = getenv("LBL_DIAG_SWEEP_BLOCKS") ? atoll(getenv("LBL_DIAG_SWEEP_BLOCKS")) : 4096;
#else
    constexpr long long cap = 4096;
#endif
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

template <bool NT, bool BUDGET>
static void launch_layer_sweep_nt(const SweepArgs& a, hipStream_t s) {
#ifdef LBL_DIAG
    static const bool pairs = !getenv("LBL_DIAG_SWEEP_NP1");
#else
    constexpr bool pairs = true;
#endif
    if (pairs && (a.first & 1) == 0 && a.count >= 2) {      // two points per thread with 16-byte accesses; an odd last point by itself
        SweepArgs m = a;
        m.count = a.count & ~1LL;
        hipLaunchKernelGGL((layer_sweep_kernel<NT, 2, BUDGET>), dim3(sweep_blocks(m.count / 2)), dim3(256), 0, s, m);
        if (a.count & 1) {
            SweepArgs t = a;
            t.first = a.first + m.count; t.count = 1;
            hipLaunchKernelGGL((layer_sweep_kernel<NT, 1, BUDGET>), dim3(1), dim3(64), 0, s, t);
        }
        return;
    }
    hipLaunchKernelGGL((layer_sweep_kernel<NT, 1, BUDGET>), dim3(sweep_blocks(a.count)), dim3(256), 0, s, a);
}

void launch_layer_sweep(const SweepArgs& a, hipStream_t s) {
    if (a.count <= 0) return;
    if (a.budget) { if (a.variant) launch_layer_sweep_nt<true, true>(a, s); else launch_layer_sweep_nt<false, true>(a, s); return; }
    if (a.variant) launch_layer_sweep_nt<true, false>(a, s);
    else launch_layer_sweep_nt<false, false>(a, s);
}

template <bool BUDGET>
static void launch_column_step_b(const ColumnStepArgs* d_args, long long first, long long count, hipStream_t s, bool kfold);

void launch_column_step(const ColumnStepArgs* d_args, long long first, long long count, hipStream_t s, int budget, int kfold) {
    if (count <= 0) return;
    if (budget) launch_column_step_b<true>(d_args, first, count, s, kfold != 0);
    else launch_column_step_b<false>(d_args, first, count, s, false);
}

template <bool BUDGET>
static void launch_column_step_b(const ColumnStepArgs* d_args, long long first, long long count, hipStream_t s, bool kfold) {
#ifdef LBL_DIAG
    static const bool pairs = !getenv("LBL_DIAG_COLUMN_NP1");
#else
    constexpr bool pairs = true;
#endif
    // default arithmetic: four points per thread (32-byte loads, three terms per batch; one Planck exp per thread and layer);
    // the reference's rounding chain: two (four measured slower there in round 3: 174 VGPRs with six terms per batch)
    if constexpr (BUDGET) {
        if (pairs && (first & 3) == 0 && count >= 4) {
            const long long quad = count & ~3LL;
            if (kfold) hipLaunchKernelGGL((column_step_kernel<4, true, true>), dim3(sweep_blocks(quad / 4)), dim3(256), 0, s, d_args, first, quad);
            else hipLaunchKernelGGL((column_step_kernel<4, true>), dim3(sweep_blocks(quad / 4)), dim3(256), 0, s, d_args, first, quad);
            if (count & 3) hipLaunchKernelGGL((column_step_kernel<1, true>), dim3(1), dim3(64), 0, s, d_args, first + quad, count & 3);
            return;
        }
    }
    if (pairs && (first & 1) == 0 && count >= 2) {     // two points per thread with 16-byte loads; an odd last point by itself
        const long long even = count & ~1LL;
        hipLaunchKernelGGL((column_step_kernel<2, BUDGET>), dim3(sweep_blocks(even / 2)), dim3(256), 0, s, d_args, first, even);
        if (count & 1) hipLaunchKernelGGL((column_step_kernel<1, BUDGET>), dim3(1), dim3(64), 0, s, d_args, first + even, 1LL);
        return;
    }
    hipLaunchKernelGGL((column_step_kernel<1, BUDGET>), dim3(sweep_blocks(count)), dim3(256), 0, s, d_args, first, count);
}

void launch_column_sweep(const ColumnArgs* d_args, long long count, hipStream_t s, int budget) {
    if (count <= 0) return;
    if (budget) hipLaunchKernelGGL(column_sweep_kernel<true>, dim3(sweep_blocks(count)), dim3(256), 0, s, d_args);
    else hipLaunchKernelGGL(column_sweep_kernel<false>, dim3(sweep_blocks(count)), dim3(256), 0, s, d_args);
}

// emissivity / absorbance / optical depth from a transmittance array
// (pyradClasses.py:73-76, 330-340, 596-606, 718-732)
__global__ __launch_bounds__(256) void optical_kernel(const double* __restrict__ tr, long long n, int kind,
                                                      double* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const double t = tr[j];
        double v;
        if (kind == 0) v = 1.0 - t;                  // emissivity = 1 - transmittance
        else if (kind == 1) v = log10(1.0 / t);      // absorbance = log10(1 / transmittance)
        else v = -log(t);                            // optical depth = -ln transmittance
        out[j] = v;
    }
}

// zeros + in[0] + in[1] + ... (pyradClasses.py:566-571, 684-689)
__global__ __launch_bounds__(256) void sum_kernel(const SumArgs A) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < A.n; j += stride) {
        double s = 0.0;
        for (int i = 0; i < A.n_in; ++i) s += A.in[i][j];
        A.out[j] = s;
    }
}

// padded all-gather result (slot r = rank r's shard) -> grid order
__global__ __launch_bounds__(256) void gather_compact_kernel(const CompactArgs A) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (int r = 0; r < A.world; ++r) {
        const double* __restrict__ src = A.gathered + (long long)r * A.slot;
        double* __restrict__ dst = A.out + A.first[r];
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < A.count[r]; i += stride) dst[i] = src[i];
    }
}

void launch_gather_compact(const CompactArgs& a, long long max_count, hipStream_t s) {
    if (max_count <= 0) return;
    hipLaunchKernelGGL(gather_compact_kernel, dim3(sweep_blocks(max_count)), dim3(256), 0, s, a);
}

void launch_sum(const SumArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(sum_kernel, dim3(sweep_blocks(a.n)), dim3(256), 0, s, a);
}

void launch_optical(const double* trans, long long n, int kind, double* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(optical_kernel, dim3(sweep_blocks(n)), dim3(256), 0, s, trans, n, kind, out);
}

void launch_planck(double* out, long long n, double start, double stop, double T, double rT, double pa, double pb, hipStream_t s) {
    if (n <= 0) return;
    const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
    hipLaunchKernelGGL(planck_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, start, stop, step,
                       T, rT, pa, pb);
}


void launch_band_integral(const double* y, long long n, double* partial, double* result, hipStream_t s) {
    const int nb = band_partial_count(n);
    hipLaunchKernelGGL(band_partial_kernel, dim3(nb), dim3(256), 0, s, y, n, partial, (long long)16384);
    hipLaunchKernelGGL(band_final_kernel, dim3(1), dim3(256), 0, s, partial, nb, result);
}

void launch_line_survey(const double* nu, const double* sw, int n_lines, double range_min, double resolution,
                        double* out, long long n_base, hipStream_t s) {
    if (n_lines <= 0) return;
    hipLaunchKernelGGL(line_survey_kernel, dim3((n_lines + 255) / 256), dim3(256), 0, s, nu, sw, n_lines, range_min,
                       resolution, out, n_base);
}

}  // namespace lbl
