// Shared host/device structures of the MI355X line-by-line engine (internal; the public
// boundary is include/pyrad_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lbl_launch_shapes.h"

namespace lbl {

// physical constants exactly as the reference spells them (pyradClasses.py:15-23,
// pyradLineshape.py:14-19, pyradIntensity.py:3-13, pyradPlanck.py:4-9)
constexpr double kB = 1.38064852E-23;
constexpr double cLight = 299792458.0;
constexpr double hPlanck = 6.62607004e-34;
constexpr double kPi = 3.141592653589793;
constexpr double kInvPi = 0.3183098861837907;            // 1/pi
constexpr double kInvSqrtPi = 0.5641895835477563;        // 1/sqrt(pi)
constexpr double t0 = 296.0;
constexpr double p0 = 1013.25;
constexpr double avo = 6.022140857E23;

// One prepared spectral line, in work-grid units, split into the part every (wave, line)
// pair reads (hot, 32 B) and the Gaussian part only pairs near the line centre read (cold,
// 32 B).  The contribution of the line to the grid point at integer offset d from its centre
// ci is
//     KL / (d*d + a2)  +  KG * exp(-b*d*d)          for |d| <= H = window-2,
// which restates pyradLineshape.py:39 (Gaussian), :52 (Lorentz) and :72-74 (pseudo-Voigt)
// times the corrected intensity of pyradIntensity.py:30-32 with x = d*resolution.
struct __attribute__((aligned(32))) HotRec {
    double cf;      // centre index (pyradClasses.py:390) as a double (exact: |ci| <= 2e9)
    double a2;      // (hw / res)^2
    double KL;      // Lorentz amplitude / res^2   (0 for a pure Gaussian line)
    int32_t dgi;    // |d| >= dgi: the Gaussian term cannot change the fp64 value of the sum
    int32_t flags;  // REC_DIRECT_DIV: denominator outside the running-fraction range
};
struct __attribute__((aligned(32))) ColdRec {
    double KG;      // Gaussian amplitude           (0 for a pure Lorentz line)
    double b;       // (res / hw)^2
    double q2;      // exp(-2 b): ratio step of the Gaussian recurrence; < 0: evaluate directly
    double KLd;     // copy of HotRec.KL for the plain-divide pass of REC_DIRECT_DIV lines
};
static_assert(sizeof(HotRec) == 32 && sizeof(ColdRec) == 32, "record halves must be 32 bytes");

enum : int32_t {
    REC_DIRECT_DIV = 1,   // Lorentz denominator outside the running-fraction range: plain divide
    REC_NO_RECUR = 2,     // Gaussian too narrow for the two-exp recurrence (b > 4): one exp per point
    REC_LONG_RUN = 4,     // Gaussian wide enough (b <= 1) for the 16-point runs of the transposed pass: a run that
                          // starts from an underflowed seed cannot reach a point where the term still matters
    REC_LONG_RUN32 = 8    // ... and for its 32-point runs (b <= 0.3)
};

// One accumulate job = one isotopologue of one layer (Isotope.createCrossSection).
// Sweep of a layer fused into the accumulate kernel's output stage (lbl_layer_step_dev): the
// arithmetic of layer_sweep_kernel; per-molecule volume fractions travel with the chain's jobs.
struct FusedSweep {
    double P, T, depth;
    double rT, r_surface_T;         // RN(1/T), RN(1/surface_T) for div_uniform (0: plain divide)
    double start, stop, step;       // xAxis = linspace(start, stop, n)
    double pa, pb, surface_T;
    const double* I_in;
    double* abs_coef; double* trans; double* I_out;
    long long n;
    int32_t on, budget;             // on: 1 the job's sum is a cross section (fold with `factor` / the IEEE chain first); 2 (merged layer job,
                                    // lbl_layer_merged_step_dev) it IS the absorption coefficient.  budget: the sweeps' default arithmetic
                                    // ("sweep_ieee_divisions" 0; see lbl_kernels.hip)
    double factor, pbkT, pbk_surface;   // for it: conc * P / 1E4 / k / T; 100 h c / k / T; 100 h c / k / surface_T
};

struct AccumJob {
    const HotRec* hot;
    const ColdRec* cold;
    const int32_t* cidx;   // centre indices, non-decreasing
    double* out;           // work grid, n_work doubles (NULL: not stored - a merged layer job whose sweep is fused in)
    int32_t n_lines;
    int32_t n_work;
    int32_t H;             // wing support in points = max(window-2, 0)
    int32_t n_tiles;
    int32_t p_begin;       // shard of the work grid computed by this job: [p_begin, p_end)
    int32_t p_end;
    int32_t flush_every;   // lines per running-fraction block: 32, or 16 for very wide windows
    int32_t pad;           // LS variant: tile order (0 XCD-chunked, 1 natural)
    int32_t span_first;    // balanced variant: first global span id of this job
    int32_t n_spans;
    // LDS variants with a host schedule: line ranges of every span of 64*R points of this job's
    // shard, 8 ints per span {iA, iB, iC, iD, iF1, iF2, 0, 0} (see wave_line_ranges[_far]); NULL:
    // the wave searches the centre indices itself
    const int32_t* span_tab;
    // Fused layer step of a single-line-list layer (lbl_layer_step_dev): the sweep of a point runs in
    // this job's output stage with the molecule's volume fraction `conc`.
    int32_t chain_flags;
    int32_t ablate;        // LBL_DIAG builds only (lbl_set_option debug_ablate, scripts/ablate.sh): timing-only runs skip parts of the kernel; else 0 and never read
    double conc;
    // Every sum leaves the kernel multiplied by out_scale: 1.0 for a line list's cross section (x * 1.0 is exact), and the
    // power of two 2^e that a merged layer job's record weights were divided by (exact as well), see PrepJob.weight.
    double out_scale;
    FusedSweep fuse;       // fuse.on: sweep every point right after its sum is final
};
enum : int32_t { CHAIN_MOL_FIRST = 1, CHAIN_MOL_LAST = 2 };

// Balanced variant: spans (64*R consecutive grid points) of all jobs of a launch group are
// numbered job-major; job j owns spans [span_first, span_first + n_spans).
struct SpanRec { int32_t iA, iB, iC, iD; };

struct PrepJob {
    const double* nu; const double* sw; const double* elower; const double* gamma_air;
    const double* gamma_self; const double* n_air; const double* delta_air;
    HotRec* hot; ColdRec* cold; int32_t* cidx;
    unsigned int* block_counts;           // [blocks of 256 lines][3]: per-block regime counts, no atomics
    double T, P, q_frac, molmass, Q_T, Q_296;
    double range_min, resolution;
    double log_t0_over_T;   // ln(296/T), computed once per job on the host
    // per-job constants of the reference's expressions, evaluated once on the host in the reference's
    // operation order (same IEEE results as on the device), and reciprocals of the per-job divisors
    double P_over_p0;       // P / p0                                   (cls:254, 258)
    double ghw_factor;      // sqrt(2 k T / m / c^2), m = molmass/1000/avo (cls:263, 296)
    double q_ratio;         // Q_296 / Q_T                              (int:30-32)
    double inv_T, inv_res, inv_res2;
    double gauss_cut;       // 2^54: the Gaussian part of a pseudo-Voigt line is evaluated until it cannot change the fp64 value of
                            // the line's sum; budget mode 2^34: until it is below 2^-34 (5.8e-11) of the line's own Lorentz term
    int32_t n_lines;
    int32_t pad;
    // Merged layer job (lbl_layer_merged_step_dev, lbl_layers_merged_accumulate_dev): the records of ALL line lists of a layer go
    // into ONE array in centre-index order, each list's amplitudes KL, KG pre-multiplied by weight = conc P / 1E4 / k / T of
    // its molecule (pyradClasses.py:583) over a power of two common to the layer, so that the accumulate kernel's sum is the
    // layer's absorption coefficient sum_m f_m sum_iso xs_iso (pyradClasses.py:707-712, 566-571) up to that exact factor.
    // merged != 0: the list belongs to a job of several lists and is prepared by that job's merged-order launch
    // (line_prep_merged_kernel), which only reads this block's constants and field pointers; weight 1.0 leaves every bit as
    // it was (x * 1.0).
    int32_t merged;
    int32_t pad2;
    double weight;
};

static_assert(sizeof(PrepJob) % 8 == 0, "line_prep_merged_kernel copies PrepJob blocks to LDS in 8-byte words");

// K1 in merged order: one per accumulate job of several line lists
struct MergedPrep {
    const int32_t* src;        // merged position -> (list within the job << 26) | line within the list
    HotRec* hot; ColdRec* cold; int32_t* cidx;       // the job's record arrays
    int32_t first_list, n_lists;                     // the job's lists in the PrepJob array
    int32_t n_total, blocks;                         // lines of all its lists; ceil(n_total / 256)
};
void launch_line_prep_merged(const PrepJob* d_lists, const MergedPrep* d_jobs, int n_jobs, int max_total, hipStream_t s);

// Merge of a layer's sorted centre-index lists (once per window, beside the schedule build): list `a` of a job holds lines
// whose centre indices (tmp_cidx, written by centre_index_kernel with K1's own expression) are non-decreasing; line i of it
// goes to  i + sum_{b < a} #{c_b <= c} + sum_{b > a} #{c_b < c}  - the stable merge, ties by list order - and what is kept is
// the inverse map of the job: src_of_job[position] = (list within the job << 26) | line.
struct MergeList {
    const double* nu;
    int32_t* tmp_cidx;     // this list's centre indices (scratch)
    int32_t* src_of_job;   // out: the job's inverse map (kept with the schedule); the same pointer in all lists of a job
    double range_min, resolution;
    int32_t n_lines;
    int32_t job_first, job_count;     // the lists [job_first, job_first + job_count) of the MergeList array form this list's job
    int32_t pad;
};
void launch_merge_ranks(const MergeList* d_lists, int n_lists, int max_lines, hipStream_t s);

// Device-side schedule build (lbl_kernels.hip "Schedule of a launch group"): one per job of the group
struct SchedJob {
    const int32_t* cidx;   // centre indices K1 wrote for this job in the current batch
    int32_t n_lines, H;
    int32_t p_begin, p_end;
    int32_t span_first;    // first span of this job in the group's span table
    int32_t tile_first;    // first (job, tile) item of this job in the group's positional item list
};
void launch_schedule_build(const SchedJob* d_jobs, int n_jobs, int total_spans, int total_tiles, int R, int spans_per_tile,
                           long long far_reach, double cost_near, double cost_edge, double cost_far, double cost_fixed,
                           int n_cu, int32_t* tabs, void* scratch, int2* worklist, hipStream_t s, int xcd_chunks = 32, bool xcd_pack = true,
                           int single_round_chunks = 1, int xcd_tol = 3, int xcd_local = -1);

// ---- fused sweep arguments (passed by value / by pointer to the sweep kernels) ----------
// line lists of one MERGED accumulate job (6 bits of the merged-order map, PrepJob blocks in K1's LDS) and arrays a layer
// sweep takes as a kernel argument; a layer with more arrays is swept by the column-step kernel, whose term list lives in
// device memory (kMaxColumnIso), so a layer holds up to kMaxColumnIso - 1 line lists (HITRAN knows ~160 isotopologues)
constexpr int kMaxIso = 64;
// A sweep walks a flat list of terms, one per cross-section array, molecule after molecule (and, in a column,
// layer after layer from the bottom): the flags say where a molecule's isotopologue sum is complete
// (pyradClasses.py:566-571 -> 583) and where a layer's molecules are (pyradClasses.py:707-716).
enum : int32_t { TERM_LAST_MOL = 1, TERM_LAST_LAYER = 2 };
struct SweepArgs {
    const double* xsec[kMaxIso];
    double term_conc[kMaxIso];      // volume fraction of the term's molecule
    int32_t term_flags[kMaxIso];
    double term_factor[kMaxIso];    // default arithmetic: conc * P / 1E4 / k / T of the term's molecule (host, the reference's order)
    double pbkT, pbk_surface;       // default arithmetic: 100 h c / k / T, 100 h c / k / surface_T
    int32_t n_iso, n_mol;
    int32_t variant, budget;         // 1: streaming (non-temporal) loads and stores (0 in LBL_DIAG builds with debug_ablate bit 64, for A/B)
    double P, T, depth;
    double rT, r_surface_T;         // RN(1/T), RN(1/surface_T) for div_uniform (0: plain divide)
    double start, stop, step;       // xAxis = linspace(start, stop, n)
    double pa, pb;                  // Planck constants 2E8*h*c**2 and 100*h*c
    double surface_T;               // used when I_in == nullptr
    const double* I_in;
    double* abs_coef; double* trans; double* I_out;
    long long n;
    long long first, count;         // swept sub-range
};
constexpr int kMaxLayers = 128;
constexpr int kMaxColumnIso = 512;      // cross-section arrays (terms) of a whole column
// Column step from the cross sections: per layer the arithmetic of layer_sweep_kernel (absorption
// coefficient, transmittance), folded bottom to top like column_sweep_kernel, in one pass.
struct ColumnStepArgs {
    const double* xsec[kMaxColumnIso];  // (a layer without line lists: one term reading the context's array of zeros)
    double term_conc[kMaxColumnIso];
    // the layer's scalars repeated per term, so that every load of a batch of terms has an address that depends
    // on the term index only (the kernel fetches them with wide scalar loads ahead of the arithmetic)
    double term_P[kMaxColumnIso], term_T[kMaxColumnIso], term_rT[kMaxColumnIso], term_depth[kMaxColumnIso];   // rT = RN(1/T) (0: plain divide)
    double term_factor[kMaxColumnIso], term_pbkT[kMaxColumnIso];     // default arithmetic (see SweepArgs)
    double pbk_surface;
    double pbkT_min, pbkT_max;          // smallest / largest term_pbkT of the column (the kernel's test for its one-exp-per-thread Planck path)
    int32_t term_flags[kMaxColumnIso];
    int32_t n_terms, n_layers;
    int32_t ablate;                     // LBL_DIAG builds only (lbl_set_option debug_ablate, scripts/ablate.sh): timing-only variants; else 0 and never read
    int32_t layer_arrays;               // any of trans[] / abs_coef[] set
    double r_surface_T;
    double* trans[kMaxLayers];          // optional per-layer transmittance outputs
    double* abs_coef[kMaxLayers];       // optional per-layer absorption coefficients
    double start, stop, step, pa, pb, surface_T;
    const double* I_in; double* I_out;
    long long n;
    long long first, count;
};
struct ColumnArgs {
    const double* trans[kMaxLayers];
    double layer_T[kMaxLayers];
    double r_layer_T[kMaxLayers];       // RN(1/layer_T[l]) (0: plain divide)
    double pbkT[kMaxLayers], pbk_surface;   // default arithmetic: 100 h c / k / T per layer and for the surface
    double r_surface_T;
    int32_t n_layers;
    double start, stop, step, pa, pb, surface_T;
    const double* I_in; double* I_out;
    long long n;
    long long first, count;
};

// ---- launchers (lbl_kernels.hip) ---------------------------------------------------------
void launch_line_prep(const PrepJob* d_jobs, int n_jobs, int max_lines, hipStream_t s);
void launch_line_quantities(const PrepJob* d_job, int n_lines, long long* index, double* lhw, double* ghw, double* intensity,
                            int32_t* regime, hipStream_t s);
void launch_accumulate(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, int LS, int variant,
                       const int2* worklist, int total_tiles, hipStream_t s, int budget = 0, int gauss_run = 16);
// narrow windows: every lane walks the lines that reach its own R points (skewed ranges); tiles of 256 R points
void launch_accumulate_skew(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, const int2* worklist, int total_tiles,
                            hipStream_t s, int LS = 1);      // LS 2 | 4: R = 8 only (waves of a workgroup share a span and deal its records)
void accumulate_far_field_params(int R, int* far_half_spans, double* far_cost, int budget = 0);
// balanced variant (4): span ranges -> prefix sum -> equal shares of (span, line) pairs per wave -> slab reduce
int balanced_workers(int R, int n_cu);
void launch_accumulate_balanced(const AccumJob* d_jobs, int n_jobs, int total_spans, int R, int n_workers,
                                SpanRec* spans, unsigned int* counts, unsigned long long* prefix, double* slab,
                                hipStream_t s);
void launch_regrid(const double* work, long long n_work, double* out, long long n_base, double start, double stop,
                   hipStream_t s);
void launch_layer_sweep(const SweepArgs& a, hipStream_t s);
// kfold: every term is a whole layer with factor 1 (the fold over absorption coefficients): the kernel then skips the per-molecule sums
void launch_column_step(const ColumnStepArgs* d_args, long long first, long long count, hipStream_t s, int budget = 0, int kfold = 0);
void launch_column_sweep(const ColumnArgs* d_args, long long n, hipStream_t s, int budget = 0);
void launch_planck(double* out, long long n, double start, double stop, double T, double rT, double pa, double pb, hipStream_t s);
void launch_band_integral(const double* y, long long n, double* partial, double* result, hipStream_t s);
struct SumArgs { const double* in[kMaxIso]; int32_t n_in; double* out; long long n; };
void launch_sum(const SumArgs& a, hipStream_t s);
constexpr int kMaxRanks = 64;
struct CompactArgs { const double* gathered; double* out; long long slot; long long first[kMaxRanks]; long long count[kMaxRanks]; int32_t world; };
void launch_gather_compact(const CompactArgs& a, long long max_count, hipStream_t s);
void launch_optical(const double* trans, long long n, int kind, double* out, hipStream_t s);
void launch_line_survey(const double* nu, const double* sw, int n_lines, double range_min, double resolution,
                        double* out, long long n_base, hipStream_t s);

}  // namespace lbl
