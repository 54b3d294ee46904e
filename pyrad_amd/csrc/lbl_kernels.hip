// Hand-written CDNA4 (gfx950) kernels of the line-by-line absorption engine.
//
//   K1 line_prep_kernel        pyradClasses.py:252-263, 378-390; pyradLineshape.py:59-71;
//                              pyradIntensity.py:16-32
//   K2 xsec_accumulate_kernel  pyradLineshape.py:39, 52, 72-74 and the scatter loop
//                              pyradClasses.py:392-400, restated as an owner-computes gather
//   K3 regrid_kernel           np.interp of pyradClasses.py:401-405 (only when res != BASE)
//   K4 layer_sweep_kernel      pyradClasses.py:566-571, 583, 707-716, 784-787; pyradPlanck.py:38-44
//   K5 column_sweep_kernel     fold of pyradClasses.py:784-787 over layers
//   K6 band_integral kernels   pyradClasses.py:26-29
//   K7 line_survey_kernel      pyradClasses.py:409-428
//
// Design notes (DESIGN.md has the long form).  The reference snaps every line centre to a
// grid index and samples the half-profile at integer multiples of the resolution, so the
// contribution of line l to grid point j depends only on |j - c_l|.  That makes the gather
// form exact: every lane owns R consecutive grid points in registers, a wavefront walks the
// (sorted) lines whose support reaches its 64*R points, line records arrive through the
// scalar cache (one s_load_dwordx16 per line, wave-uniform) and each grid point is written
// once with a plain coalesced store.  No atomics, no LDS traffic in the inner loop, and a
// fixed summation order (line order) per grid point, so two runs are bit-identical.
//
// The kernel is fp64-VALU bound (about 5 fp64 instructions per line x grid-point pair), not
// HBM bound: its compulsory traffic is 64 B per line and 8 B per grid point.
#include "lbl_device.h"

namespace lbl {

// ----------------------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Line records are written by K1 and only read by K2: viewing them through the constant
// address space tells the compiler the memory is invariant for the kernel, so a wave-uniform
// index turns into one s_load_dwordx16 (scalar cache -> SGPRs) instead of 64 lanes of flat loads.
typedef const double __attribute__((address_space(4)))* RecPtr;      // 8 doubles per record
typedef const int32_t __attribute__((address_space(4)))* RecIntPtr;
__device__ __forceinline__ RecPtr as_const_recs(const LineRec* p) {
    return (RecPtr)(unsigned long long)p;
}
__device__ __forceinline__ LineRec load_rec(RecPtr recs, int i) {
    RecPtr p = recs + (long long)i * 8;
    RecIntPtr q = (RecIntPtr)(p + 6);
    LineRec r;
    r.cf = p[0]; r.a2 = p[1]; r.KL = p[2]; r.KG = p[3]; r.b = p[4]; r.q2 = p[5];
    r.ci = q[0]; r.dgi = q[1]; r.flags = q[2]; r.pad = 0;
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
__device__ __forceinline__ double planck_wn(double n, double T, double pa, double pb) {
    double a = pa * (n * n * n);
    double b = pb * n / kB / T;
    return a / (exp(b) - 1.0);
}

// ----------------------------------------------------------------------------------------
// K1: per-line preparation
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void line_prep_kernel(const PrepJob* __restrict__ jobs) {
    const PrepJob& J = jobs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int regime = -1;
    if (i < J.n_lines) {
        const double nu = J.nu[i];
        const double T = J.T, P = J.P, q = J.q_frac;
        // Line.broadenedLine (pyradClasses.py:252-254)
        const double broadened = nu + J.delta_air[i] * P / p0;
        // Line.lorentzHW (pyradClasses.py:256-259)
        const double lhw = ((1.0 - q) * J.gamma_air[i] + q * J.gamma_self[i]) * (P / p0) * pow(t0 / T, J.n_air[i]);
        // Isotope.molMass (pyradClasses.py:294-296), Line.gaussianHW (pyradClasses.py:261-263)
        const double m = J.molmass / 1000.0 / avo;
        const double ghw = broadened * sqrt(2.0 * kB * T / m / (cLight * cLight));
        const double ratio = lhw / ghw;                                   // pyradClasses.py:378
        // pyradIntensity.intensityFactor (pyradIntensity.py:16-32) at the SHIFTED wavenumber
        // (pyradClasses.py:388)
        const double c2 = cLight * hPlanck * 100.0 / kB;                  // pyradIntensity.py:13
        const double E = J.elower[i];
        const double stim = (1.0 - exp(-c2 * broadened / T)) / (1.0 - exp(-c2 * broadened / t0));
        const double boltz = exp(-c2 * E / T) / exp(-c2 * E / t0);
        const double A = J.sw[i] * (J.Q_296 / J.Q_T) * stim * boltz;
        // centre index from the UNSHIFTED wavenumber, truncation toward zero (pyradClasses.py:390)
        const double fidx = (nu - J.range_min) / J.resolution;
        long long idx = (long long)fidx;
        if (idx > 2000000000LL) idx = 2000000000LL;
        if (idx < -2000000000LL) idx = -2000000000LL;

        const double res = J.resolution;
        double hw, KL, KG;
        if (ratio < .01) {                // Gaussian only (pyradClasses.py:379-381)
            regime = 0;
            hw = ghw;
            KL = 0.0;
            KG = A / hw / sqrt(kPi);                                      // pyradLineshape.py:39
        } else if (ratio > 100.0) {       // Lorentz only (pyradClasses.py:382-384)
            regime = 1;
            hw = lhw;
            KL = A * (hw / kPi);                                          // pyradLineshape.py:52
            KG = 0.0;
        } else {                          // pseudo-Voigt (pyradClasses.py:385-387, pyradLineshape.py:58-76)
            regime = 2;
            const double g = 2.0 * ghw, l = 2.0 * lhw;
            const double g2 = g * g, l2 = l * l;
            const double f5 = g2 * g2 * g + 2.69269 * g2 * g2 * l + 2.42843 * g2 * g * l2 +
                              4.47163 * g2 * l2 * l + .07842 * g * l2 * l2 + l2 * l2 * l;
            const double f = pow(f5, .2);
            const double x = l / f;
            const double eta = 1.36603 * x - .47719 * x * x + .11116 * x * x * x;
            hw = f / 2.0;
            KL = eta * (A * (hw / kPi));
            KG = (1.0 - eta) * (A / hw / sqrt(kPi));
        }
        const double a = hw / res;
        LineRec r;
        r.cf = (double)idx;
        r.ci = (int32_t)idx;
        r.a2 = a * a;
        r.KL = KL / (res * res);
        r.KG = KG;
        r.b = 1.0 / r.a2;
        r.flags = 0;
        r.pad = 0;
        // running-fraction accumulation multiplies up to 16 denominators (< 4.7e18 + a2):
        // keep lines whose a2 could over/underflow that product on the plain-divide path
        if (!(r.a2 > 1e-16 && r.a2 < 1e16)) r.flags |= REC_DIRECT_DIV;
        // Gaussian term: where can it still change the fp64 value of the line's sum?
        double dg = 0.0;
        if (KG != 0.0) {
            const double u2_under = 745.2;           // exp(-745.2) == 0 in fp64
            double u2 = u2_under;
            if (KL != 0.0) {
                // ratio Gauss/Lorentz at offset u = d/a:  C (1+u^2) exp(-u^2);  solve = 2^-54
                const double C = fabs(KG / (r.KL * r.b)) * 18014398509481984.0;
                if (C <= 1.0) {
                    u2 = 0.0;
                } else {
                    double v = log(C);
                    for (int it = 0; it < 6; ++it) v = log(C * (1.0 + v));
                    u2 = fmin(v + 1.0, u2_under);
                }
            }
            dg = (u2 > 0.0) ? sqrt(u2) * a + 2.0 : 0.0;
        }
        r.dgi = (dg < 2.0e9) ? (int32_t)dg : 2000000000;
        // Gaussian recurrence along a lane's consecutive points needs exp(b*R^2) finite and
        // well inside the normal range; very narrow profiles take the direct exp instead
        r.q2 = (r.b <= 8.0) ? exp(-2.0 * r.b) : -1.0;
        J.recs[i] = r;
        J.cidx[i] = r.ci;
        if (J.dbg_index) J.dbg_index[i] = (long long)fidx;
        if (J.dbg_lhw) J.dbg_lhw[i] = lhw;
        if (J.dbg_ghw) J.dbg_ghw[i] = ghw;
        if (J.dbg_intensity) J.dbg_intensity[i] = A;
        if (J.dbg_regime) J.dbg_regime[i] = regime;
    }
    // regime counters (pyradClasses.py:368-370, 406): one atomic per wave and regime
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < 3; ++k) {
        const unsigned long long m = __ballot(regime == k);
        if (lane == 0 && m) atomicAdd(&J.regime_counts[k], (unsigned long long)__popcll(m));
    }
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

// Gaussian part of one line for a lane's R consecutive points.
//   GM == 0: one exp per point.
//   GM == 1: two exps per lane, then g(d+1) = g(d) r(d), r(d+1) = r(d) q2 walking AWAY from
//            the centre (lanes left of the centre are mirrored so the terms only decay).
template <int R, bool MASKED, int GM>
__device__ __forceinline__ void gauss_term(const LineRec& r, double d0, double Hf, double (&acc)[R]) {
    if (GM == 0 || r.q2 < 0.0 || R < 4) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            double t = r.KG * exp(-r.b * (d * d));
            if (MASKED) t = (fabs(d) <= Hf) ? t : 0.0;
            acc[k] += t;
        }
    } else {
        const bool mirror = (2.0 * d0 + (double)(R - 1)) < 0.0;
        const double e0 = mirror ? -(d0 + (double)(R - 1)) : d0;
        double g = r.KG * exp(-r.b * (e0 * e0));
        double rr = exp(-r.b * (2.0 * e0 + 1.0));
        double t[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            t[k] = g;
            g *= rr;
            rr *= r.q2;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            double v = mirror ? t[R - 1 - k] : t[k];
            if (MASKED) {
                const double d = d0 + (double)k;
                v = (fabs(d) <= Hf) ? v : 0.0;
            }
            acc[k] += v;
        }
    }
}

// Lines [i0, i1) against this lane's R points.  MASKED: test |d| <= H per point (lines whose
// support ends inside the wave's span); otherwise every point of the wave is inside.
//   DIV == 0: one IEEE divide per pair.
//   DIV == 2: running fraction N/D over blocks of 16 lines (N <- N den + K D, D <- D den),
//             one divide per point and block.
template <int R, bool MASKED, int DIV, int GM>
__device__ __forceinline__ void process_lines(RecPtr recs, int i0, int i1,
                                              double x0, int wlo, int whi, double Hf, double (&acc)[R]) {
    if (DIV == 0) {
        for (int i = i0; i < i1; ++i) {
            const LineRec r = load_rec(recs, i);
            const double d0 = x0 - r.cf;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const double d = d0 + (double)k;
                const double den = fma(d, d, r.a2);
                double t = r.KL / den;
                if (MASKED) t = (fabs(d) <= Hf) ? t : 0.0;
                acc[k] += t;
            }
            const int dist = max(0, max(r.ci - whi, wlo - r.ci));
            if (dist < r.dgi) gauss_term<R, MASKED, GM>(r, d0, Hf, acc);
        }
    } else {
        constexpr int F = 16;
        for (int ib = i0; ib < i1; ib += F) {
            const int ie = min(ib + F, i1);
            double N[R], D[R];
#pragma unroll
            for (int k = 0; k < R; ++k) { N[k] = 0.0; D[k] = 1.0; }
            for (int i = ib; i < ie; ++i) {
                const LineRec r = load_rec(recs, i);
                const double d0 = x0 - r.cf;
                if (r.flags & REC_DIRECT_DIV) {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double d = d0 + (double)k;
                        double t = r.KL / fma(d, d, r.a2);
                        if (MASKED) t = (fabs(d) <= Hf) ? t : 0.0;
                        acc[k] += t;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        const double d = d0 + (double)k;
                        const double den = fma(d, d, r.a2);
                        double K = r.KL;
                        if (MASKED) K = (fabs(d) <= Hf) ? K : 0.0;
                        const double t = K * D[k];
                        N[k] = fma(N[k], den, t);
                        D[k] *= den;
                    }
                }
                const int dist = max(0, max(r.ci - whi, wlo - r.ci));
                if (dist < r.dgi) gauss_term<R, MASKED, GM>(r, d0, Hf, acc);
            }
#pragma unroll
            for (int k = 0; k < R; ++k) acc[k] += N[k] / D[k];
        }
    }
}

template <int R, int DIV, int GM>
__global__ __launch_bounds__(256) void xsec_accumulate_kernel(const AccumJob* __restrict__ jobs) {
    const AccumJob& J = jobs[blockIdx.y];
    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
    // contiguous run of tiles: neighbouring tiles read almost the same line records.
    const int n_tiles = J.n_tiles;
    const int chunk = (n_tiles + 7) >> 3;
    const int b = blockIdx.x;
    const int slot = b >> 3;
    if (slot >= chunk) return;
    const int tile = (b & 7) * chunk + slot;
    if (tile >= n_tiles) return;

    const int lane = threadIdx.x & 63;
    const int wave = uniform_i32(threadIdx.x >> 6);
    const int n_work = J.p_end;          // this job computes work-grid points [p_begin, p_end)
    const long long wave_lo_ll = (long long)J.p_begin + (long long)tile * (256LL * R) + (long long)wave * (64LL * R);
    if (wave_lo_ll >= n_work) return;
    const int wlo = (int)wave_lo_ll;
    const int whi = min(wlo + 64 * R - 1, n_work - 1);
    const int H = J.H;

    // line ranges of this wave (cidx is sorted):
    //   [iA, iB)  left-edge lines,   c in [wlo-H, whi-H)      -> masked
    //   [iB, iC)  interior lines,    c in [whi-H, wlo+H]      -> every point inside the support
    //   [iC, iD)  right-edge lines,  c in (wlo+H, whi+H]      -> masked
    int iA, iB, iC, iD;
    {
        const long long tA = (long long)wlo - H, tB = (long long)whi - H;
        const long long tC = (long long)wlo + H + 1, tD = (long long)whi + H + 1;
        const long long tgt = lane == 0 ? tA : lane == 1 ? tB : lane == 2 ? tC : tD;
        const int pos = lower_bound_i32(J.cidx, J.n_lines, tgt);
        iA = __builtin_amdgcn_readlane(pos, 0);
        iB = __builtin_amdgcn_readlane(pos, 1);
        iC = __builtin_amdgcn_readlane(pos, 2);
        iD = __builtin_amdgcn_readlane(pos, 3);
        if (tB >= tC) { iB = iD; iC = iD; }      // span wider than the support: no interior line
    }

    const int p0 = wlo + lane * R;
    const double x0 = (double)p0;
    const double Hf = (double)H;
    double acc[R];
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0.0;

    const RecPtr recs = as_const_recs(J.recs);
    process_lines<R, true, DIV, GM>(recs, iA, iB, x0, wlo, whi, Hf, acc);
    process_lines<R, false, DIV, GM>(recs, iB, iC, x0, wlo, whi, Hf, acc);
    process_lines<R, true, DIV, GM>(recs, iC, iD, x0, wlo, whi, Hf, acc);

    double* __restrict__ out = J.out;
    if (p0 + R <= n_work) {
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
            if (p0 + k < n_work) out[p0 + k] = acc[k];
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
__global__ __launch_bounds__(256) void layer_sweep_kernel(const SweepArgs A) {
#pragma clang fp contract(off)
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long jend = A.first + A.count;
    for (long long j = A.first + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < jend; j += stride) {
        // Layer.absCoef (pyradClasses.py:707-712): zeros + sum over molecules of
        // Molecule.absCoef = crossSection * concentration * P / 1E4 / k / T (pyradClasses.py:583),
        // Molecule.crossSection = zeros + sum over isotopologues (pyradClasses.py:566-571)
        double kk = 0.0;
        int i = 0;
        for (int m = 0; m < A.n_mol; ++m) {
            double xs = 0.0;
            while (i < A.n_iso && A.iso_mol[i] == m) { xs += A.xsec[i][j]; ++i; }
            kk += xs * A.conc[m] * A.P / 1E4 / kB / A.T;
        }
        if (A.abs_coef) A.abs_coef[j] = kk;
        const double tr = exp(-kk * A.depth);                               // pyradClasses.py:716
        if (A.trans) A.trans[j] = tr;
        if (A.I_out) {
            const double nu = linspace_at(j, A.n, A.start, A.stop, A.step);
            const double B = planck_wn(nu, A.T, A.pa, A.pb);                // Layer.planck(self.T)
            const double Iin = A.I_in ? A.I_in[j] : planck_wn(nu, A.surface_T, A.pa, A.pb);
            const double transmitted = tr * Iin;                            // pyradClasses.py:785
            const double emitted = (1.0 - tr) * B;                          // pyradClasses.py:786
            A.I_out[j] = transmitted + emitted;
        }
    }
}

// ----------------------------------------------------------------------------------------
// K5: column fold (pyradClasses.py:784-787 applied layer after layer)
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void column_sweep_kernel(const ColumnArgs* __restrict__ Ap) {
#pragma clang fp contract(off)
    const ColumnArgs& A = *Ap;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long jend = A.first + A.count;
    for (long long j = A.first + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < jend; j += stride) {
        const double nu = linspace_at(j, A.n, A.start, A.stop, A.step);
        double I = A.I_in ? A.I_in[j] : planck_wn(nu, A.surface_T, A.pa, A.pb);
        for (int l = 0; l < A.n_layers; ++l) {
            const double tr = A.trans[l][j];
            const double B = planck_wn(nu, A.layer_T[l], A.pa, A.pb);
            const double transmitted = tr * I;
            const double emitted = (1.0 - tr) * B;
            I = transmitted + emitted;
        }
        A.I_out[j] = I;
    }
}

__global__ __launch_bounds__(256) void planck_kernel(double* __restrict__ out, long long n, double start,
                                                     double stop, double step, double T, double pa, double pb) {
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = planck_wn(linspace_at(j, n, start, stop, step), T, pa, pb);
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

template <int R>
static void launch_accum_r(const AccumJob* d_jobs, int n_jobs, int max_tiles, int variant, hipStream_t s) {
    dim3 grid(((max_tiles + 7) / 8) * 8, n_jobs);
    switch (variant) {
        case 0: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 0, 0>), grid, dim3(256), 0, s, d_jobs); break;
        case 1: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 2, 0>), grid, dim3(256), 0, s, d_jobs); break;
        default: hipLaunchKernelGGL((xsec_accumulate_kernel<R, 2, 1>), grid, dim3(256), 0, s, d_jobs); break;
    }
}

void launch_accumulate(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, int variant, hipStream_t s) {
    if (n_jobs <= 0 || max_tiles <= 0) return;
    switch (R) {
        case 1: launch_accum_r<1>(d_jobs, n_jobs, max_tiles, variant, s); break;
        case 2: launch_accum_r<2>(d_jobs, n_jobs, max_tiles, variant, s); break;
        case 4: launch_accum_r<4>(d_jobs, n_jobs, max_tiles, variant, s); break;
        default: launch_accum_r<8>(d_jobs, n_jobs, max_tiles, variant, s); break;
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

static int sweep_blocks(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

void launch_layer_sweep(const SweepArgs& a, hipStream_t s) {
    if (a.count <= 0) return;
    hipLaunchKernelGGL(layer_sweep_kernel, dim3(sweep_blocks(a.count)), dim3(256), 0, s, a);
}

void launch_column_sweep(const ColumnArgs* d_args, long long count, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(column_sweep_kernel, dim3(sweep_blocks(count)), dim3(256), 0, s, d_args);
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

void launch_sum(const SumArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(sum_kernel, dim3(sweep_blocks(a.n)), dim3(256), 0, s, a);
}

void launch_optical(const double* trans, long long n, int kind, double* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(optical_kernel, dim3(sweep_blocks(n)), dim3(256), 0, s, trans, n, kind, out);
}

void launch_planck(double* out, long long n, double start, double stop, double T, double pa, double pb, hipStream_t s) {
    if (n <= 0) return;
    const double step = n > 1 ? (stop - start) / (double)(n - 1) : 0.0;
    hipLaunchKernelGGL(planck_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, start, stop, step,
                       T, pa, pb);
}

int band_partial_count(long long n) {
    long long b = (n + 16383) / 16384;       // 16384 points per block
    return (int)(b < 1 ? 1 : b);
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
