/*
 * pyrad_hip.h — C ABI of the MI355X (gfx950) line-by-line absorption engine.
 *
 * The reference (bschrag620/PyRad) has no FFI: its seam is a Python method contract,
 * Isotope.createCrossSection() (pyradClasses.py:361-407) plus the property chain
 * absCoef -> transmittance -> transmission (pyradClasses.py:322-340, 581-606, 707-732,
 * 784-787) and pyradPlanck.planckWavenumber (pyradPlanck.py:38-44).  This header is the
 * C boundary a maintainer binds with ctypes in place of those bodies.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns int
 *     (LBL_OK or a negative lbl_status); no C++ exception crosses the boundary.
 *   - lbl_last_error(ctx) returns a NUL-terminated description of the last failure
 *     on that context (ctx == NULL: last failure of a ctx-less call on this thread).
 *   - Host pointers are C-contiguous float64 arrays owned by the caller; the library
 *     never keeps a host pointer past return.
 *   - Device objects (lbl_lines, lbl_buffer, lbl_comm) belong to the context that made
 *     them and must be destroyed before it.
 *   - A context is bound to one HIP device and one HIP stream and is NOT thread-safe.
 *     "_dev" entry points only enqueue work on the context stream; lbl_sync() or any
 *     download drains it.  Host-pointer entry points are synchronous.
 *   - All arithmetic is IEEE fp64 (the reference is NumPy float64); grid indices int64.
 */
#ifndef PYRAD_HIP_H
#define PYRAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBL_ABI_VERSION 5

typedef enum lbl_status {
    LBL_OK = 0,
    LBL_ERR_BAD_ARG = -1,     /* NULL pointer, negative size, unsorted lines, ... */
    LBL_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    LBL_ERR_HIP = -3,         /* a HIP runtime call failed */
    LBL_ERR_RCCL = -4,        /* an RCCL call failed */
    LBL_ERR_OOM = -5,         /* device allocation failed */
    LBL_ERR_STATE = -6        /* object belongs to another context, comm not initialised, ... */
} lbl_status;

typedef struct lbl_ctx lbl_ctx;
typedef struct lbl_lines lbl_lines;     /* device-resident HITRAN line list (SoA, sorted by nu) */
typedef struct lbl_buffer lbl_buffer;   /* device-resident float64 array */
typedef struct lbl_comm lbl_comm;       /* RCCL communicator, one rank per process */

/* Per-isotopologue scalars read by Isotope.createCrossSection (pyradClasses.py:361-407):
 * layer.T, layer.P (mbar), molecule.concentration (volume fraction, the q of
 * Line.lorentzHW, pyradClasses.py:258), isotope.molmass (g/mol, pyradClasses.py:296),
 * isotope.q[layer.T] and isotope.q296 (pyradClasses.py:389). */
typedef struct lbl_iso_params {
    double T;
    double P;
    double q_frac;
    double molmass;
    double Q_T;
    double Q_296;
} lbl_iso_params;

/* Layer grid (pyradClasses.py:648-676, 698-705).  The host computes these with the
 * reference's own expressions so that the integer truncations agree:
 *   resolution      = layer.resolution                        (pyradClasses.py:659-662)
 *   n_work          = int((rangeMax-rangeMin)/resolution)      (pyradClasses.py:700)
 *   n_base          = int((rangeMax-rangeMin)/BASE_RESOLUTION) (pyradClasses.py:672)
 *   window          = len(arange(0, distanceFromCenter, resolution))  (pyradClasses.py:377)
 * Wing support of a line is centre +- (window-2) grid points (pyradClasses.py:394). */
typedef struct lbl_grid {
    double range_min;
    double range_max;
    double resolution;
    double base_resolution;
    int64_t n_work;
    int64_t n_base;
    int64_t window;
    /* Contiguous shard of the WORK grid this call computes: points
     * [shard_first, shard_first + shard_count).  shard_count == 0 means the whole grid.
     * Output buffers stay globally indexed (n_base long) so that the shards of all ranks
     * can be all-gathered in place.  A sharded call requires resolution == base_resolution. */
    int64_t shard_first;
    int64_t shard_count;
} lbl_grid;

/* ---- library / context ------------------------------------------------------------- */
int lbl_abi_version(void);
/* Fixed sizes of the library (ABI 5), so that a host can choose a route instead of running into LBL_ERR_BAD_ARG:
 *   "merged_lists_per_job"  line lists one merged layer job takes (lbl_layer_merged_step_dev, lbl_layers_merged_accumulate_dev): 64
 *   "arrays_per_layer"      cross-section arrays of lbl_layer_sweep_dev / line lists of lbl_layer_step_dev: 511 (the reference
 *                           sums however many isotopologues a layer holds, pyradClasses.py:566-571, 707-712; HITRAN has ~160)
 *   "arrays_per_sum"        inputs of lbl_sum_dev: 64 (a longer sum is chained: the partial sum first)
 *   "arrays_per_column"     terms of lbl_column_step_dev: 511      "layers_per_column": 128      "jobs_per_batch": LBL_MAX_JOBS
 * Unknown name: LBL_ERR_BAD_ARG. */
int lbl_limit(const char* name, int64_t* value);
int lbl_device_count(int* count);
int lbl_ctx_create(int device, lbl_ctx** out);
int lbl_ctx_destroy(lbl_ctx* ctx);
const char* lbl_last_error(const lbl_ctx* ctx);
int lbl_sync(lbl_ctx* ctx);
/* Native hipStream_t of the context (as void*) so a host can order foreign work on it. */
int lbl_ctx_stream(lbl_ctx* ctx, void** stream);
/* Two contexts of one device as a software pipeline: the ACCUMULATE kernels of `ctx` wait for the accumulate
 * kernels `predecessor` has enqueued so far (an event on its stream), everything else of `ctx` - line prep
 * before them, the sweep after them - does not.  With A chained after B and B after A and steps dealt
 * alternately, the fp64-bound accumulate kernels run one after another, each alone on the chip, while the
 * line prep of the next step and the sweep of the previous one fill the cycles they leave; every step's
 * arrays are complete in step order.  predecessor = NULL ends the chaining.
 * Destroying a predecessor unlinks it: its successors simply stop waiting (any destroy order is safe).
 * A chained context (either end of a link) cannot capture a graph - the cross-context event waits do not
 * live in a captured sequence: lbl_capture_begin returns LBL_ERR_STATE on it, and a capturing context
 * cannot be chained. */
int lbl_ctx_chain_accumulate(lbl_ctx* ctx, lbl_ctx* predecessor);
/* Name of the device ("gfx950..."), CU count, HBM bytes. */
int lbl_device_info(lbl_ctx* ctx, char* name, int name_len, int* n_cu, int64_t* hbm_bytes);

/* Tuning knobs for A/B parity runs and benchmarking (no reference counterpart):
 *   "accum_variant"          0 the literal form: IEEE divide + exp per (line, grid point) pair, line records through
 *                            the scalar cache (slow; the on-device cross-check of the others) |
 *                            3 running fraction + Gaussian recurrence with wave-private LDS staging of the
 *                            records, every pair evaluated directly |
 *                            5 (default) = 3 with the fp64-exact far-field series for Lorentz lines
 *                            more than 4 half-spans away from a span of 64*R points.
 *                            (1, 2 and 4 - superseded comparison kernels - exist in diagnostic builds of the
 *                            library only, make EXTRA=-DLBL_DIAG; the production library answers LBL_ERR_BAD_ARG)
 *   "accum_points_per_lane"  0 (auto) | 1 | 2 | 4 | 8
 *   "accum_line_split"       0 (auto) | 1 | 2 | 4 | 8 waves of a workgroup share one span of points
 *                            and split its lines (variants 3 and 5)
 *   "accum_longest_first"    workgroups are dispatched from a cached (job, tile) worklist sorted by
 *                            decreasing cost: 3 bin-packed per CU when the launch is a single round of
 *                            workgroups, 2 every other tier of n_cu items reversed (snake), 1 plain |
 *                            0: positional order (waves then search their line ranges themselves) |
 *                            4 (default): as 3, and a launch of several rounds is XCD-partitioned: workgroup
 *                            i runs on XCD i mod 8, each XCD has its own L2, so every XCD gets 32 contiguous
 *                            chunks of the tile sequence (dealt round-robin, equal estimated cost), each
 *                            XCD's list longest-first - K2 fetches 33 MB instead of 110 MB on the
 *                            100-2500 cm^-1 cell at the same kernel time
 *   "accum_tile_order"       positional order only: 1 (default) natural | 0 one contiguous run of
 *                            tiles per XCD | 2 golden-ratio stride
 *   "accum_blocks_per_cu"    variant 4 (diagnostic builds) only: resident workgroups per CU, 0 = ask the runtime
 *   "accum_skew"             1 (default) line lists whose window has no far line (narrower than 640 points) go
 *                            through the skewed-range kernel when they fill the chip | 0 the span kernel
 *                            (all-direct instantiation) | 2 EVERY job through the skewed-range kernel (parity tests)
 *   "accum_skew_points_per_lane"  1 | 2 | 4 | 8 (default)
 *   "accum_xcd_chunks"       XCD-partitioned order ("accum_longest_first" 4): contiguous chunks of the tile sequence per
 *                            XCD, 0 (default: about 29 workgroups per chunk, 10..32 chunks) .. 64; for a launch of one
 *                            round: the contiguous runs per XCD of "accum_xcd_pack" (default 1, at most 16)
 *   "accum_xcd_pack"         launches of one round (at most 4 workgroups per CU; device-built schedules): every XCD packs its own
 *                            tiles into its own CUs.  1 (default): where every wave owns a span (no line split) an XCD's tiles
 *                            are a contiguous run of the sequence worth an eighth of the cost (its L2 then holds that run's
 *                            records only), else every 8th tile of the longest-first order (waves that share spans AND are
 *                            neighbours in the spectrum queue for the same L2 lines: +9..17 % measured) | 2 always the run |
 *                            3 always the mix | 0 one packing over all CUs by one wave (round 4).  The dispatch list has idle
 *                            positions (workgroups that exit at once)
 *   "accum_xcd_tolerance"    a contiguous run is kept as long as the busiest CU of no XCD carries more than this many percent
 *                            (default 3; -1: any) above the mean of the eight by the cost model: an XCD whose run holds the
 *                            spectrum's expensive tiles cannot hand any to another XCD's CUs; the mix takes over
 *   "accum_skew_line_split"  0 (default: by the lines per grid point) | 1 | 2 | 4 waves of a workgroup share one span of the
 *                            skewed-range kernel and deal its records (dense, merged line lists: a chunk of records
 *                            then covers the span again)
 *   "accum_gauss_run"        far-field kernel, production shape (4 points per lane, unsplit spans): points a lane walks per Gaussian
 *                            run.  16: two exp per 16 points, 128 VGPRs, four waves per SIMD | 32: two exp per 32 points, 164-166
 *                            VGPRs, three waves per SIMD; in exact mode also the build whose far-field series starts at 3
 *                            half-spans instead of 4 (38 terms) | 0 (default): budget mode - 32 for launches of more than 16 waves
 *                            per SIMD, else 16; exact mode - 32 except for launches of 12-16 waves per SIMD (one round of the
 *                            chip's wave slots at four per SIMD, two at three).  Results of the two builds agree to ~1e-15
 *                            (different summation orders), each is reproducible bit for bit.
 *   "accum_far_min_window"   windows below this many points take the skewed-range kernel even where the far-field
 *                            kernel could run them (0, the default: its own limit, 640; measured flat up to 1000)
 *   "debug_ablate"           ONLY in diagnostic builds of the library (make EXTRA=-DLBL_DIAG): timing experiments,
 *                            bits switch off parts of kernels, results are wrong.  The production library has no
 *                            such code in its kernels and answers LBL_ERR_BAD_ARG (unknown option)
 *   "accuracy"               0 (default) "exact": every array as close to the reference's fp64 values as the arithmetic allows
 *                            (measured 1e-14 at every grid point of every BASELINE configuration) |
 *                            1 "budget": <= 1e-9 relative on the absorption coefficient (BASELINE north_star asks for 1e-6),
 *                            everything still fp64: 18 / 12 / 9 / 7 far-field series terms by distance instead of
 *                            30 / 20 / 15 / 12 (remainder <= 5.9e-10 of a line's own term), the Gaussian part of a pseudo-Voigt
 *                            line dropped where it is below 2^-34 of the line's Lorentz part (exact: 2^-54).  Applies to the
 *                            batches enqueued after the call
 *   "sweep_ieee_divisions"   0 (default) the sweeps form the absorption coefficient as cross section x one host-computed
 *                            factor conc P / 1E4 / k / T, the Planck exponent as n x (100 h c / k / T), reciprocals by
 *                            rcp + Newton steps: a few 1e-16 from | 1 the reference's own chain of correctly rounded
 *                            divisions (k bit-identical to NumPy's crossSection * concentration * P / 1E4 / k / T on the
 *                            same cross section; 2.5x the instructions per point and layer)
 *   "schedule_build"         1 (default) span tables and dispatch order of a launch group are built on the device, in
 *                            stream, by the first batch that uses them (no host search, no copy, no wait) | 0 on the host
 *                            (one thread; 4 ms for the 100-2500 cm^-1 cell, 80 ms for a 30-layer column).  Same tables,
 *                            same results; other "accum_longest_first" values than 4 always build on the host
 *   "layer_step_fused"       1 (default) lbl_layer_step_dev folds the sweep of a single-line-list layer into the
 *                            accumulate kernel | 0 always accumulate launch + sweep launch (bit-identical; A/B)
 *   "debug_throw"            test hook: 1 / 2 / 3 raise std::bad_alloc / std::runtime_error /
 *                            std::length_error inside the library; the call must come back as
 *                            LBL_ERR_OOM / LBL_ERR_STATE / LBL_ERR_OOM (no exception crosses this boundary) */
int lbl_set_option(lbl_ctx* ctx, const char* key, int value);

/* Kernel timing with HIP events recorded on the context stream around every launch of a
 * kernel class (the stream the kernels run on; torch.cuda.Event would not see it).
 * kind: 0 line_prep, 1 xsec_accumulate, 2 regrid, 3 layer_sweep, 4 column_sweep, 5 all-gather.
 * `on` is a bit mask of kinds (bit k = kind k; 0x3F = all; <= 0 = off).
 * lbl_profile_read drains the stream, returns the number of launches recorded since the last
 * reset and their summed duration in milliseconds. */
int lbl_profile_enable(lbl_ctx* ctx, int on);
int lbl_profile_read(lbl_ctx* ctx, int kind, int64_t* launches, double* total_ms);
int lbl_profile_reset(lbl_ctx* ctx);
/* Create events ahead of time so that timed launches only record them (an event created on first
 * use costs the timed region ~10 us). */
int lbl_profile_reserve(lbl_ctx* ctx, int n_events);

/* ---- device buffers (float64) ------------------------------------------------------- */
int lbl_buffer_create(lbl_ctx* ctx, int64_t n, lbl_buffer** out);
int lbl_buffer_destroy(lbl_buffer* buf);
int lbl_buffer_size(const lbl_buffer* buf, int64_t* n);
int lbl_buffer_upload(lbl_buffer* buf, const double* host, int64_t n, int64_t dst_offset);
int lbl_buffer_download(lbl_buffer* buf, double* host, int64_t n, int64_t src_offset);
int lbl_buffer_fill(lbl_buffer* buf, double value);                    /* async */
/* Download that does not wait: ordered behind everything enqueued on the context stream so far, carried out on the
 * context's copy stream beside the kernels enqueued after it (pyrad_amd.model sends a column's outgoing spectrum -
 * Atmosphere.transmission, pyradClasses.py:784-787 over all layers - home in pieces while the next piece is folded).
 * `host` should be page-locked (lbl_host_alloc); the range must not be rewritten, nor `host` read, before
 * lbl_download_wait (lbl_sync waits for these copies too).  Not capturable. */
int lbl_buffer_download_async(lbl_buffer* buf, double* host, int64_t n, int64_t src_offset);
int lbl_download_wait(lbl_ctx* ctx);
/* Page-locked host memory for the arrays a caller keeps handing to upload / download: the copy then
 * runs at the link's DMA rate instead of through the runtime's staging of pageable memory (3-5x
 * faster for a spectrum of a few MB).  Plain memory otherwise; free it with lbl_host_free (ctx may be
 * NULL there if the context is already gone). */
int lbl_host_alloc(lbl_ctx* ctx, int64_t bytes, void** out);
int lbl_host_free(lbl_ctx* ctx, void* ptr);
int lbl_buffer_devptr(lbl_buffer* buf, void** devptr);                 /* for RCCL / interop */

/* ---- line lists ------------------------------------------------------------------- */
/* Upload the seven per-line HITRAN fields Line carries and uses (pyradClasses.py:237-263;
 * Einstein A is carried by the reference but never read).  nu must be non-decreasing. */
int lbl_lines_create(lbl_ctx* ctx, const double* nu, const double* sw, const double* elower,
                     const double* gamma_air, const double* gamma_self, const double* n_air,
                     const double* delta_air, int64_t n_lines, lbl_lines** out);
/* A view of `count` consecutive lines of a resident list, starting at line `first` (nu is sorted, so a wavenumber
 * window of a list - Layer.effectiveRangeMin/Max of pyradClasses.py:656-657, or the halo of a grid shard - is such a
 * range): no copy, the view shares the parent's device arrays.  A 30-layer column keeps ONE copy of every molecule's
 * lines and 30 views of it, and the line-prep kernels of all layers read the same addresses (they stay in cache).
 * A view is destroyed with lbl_lines_destroy; a list with live views cannot be destroyed (LBL_ERR_STATE). */
int lbl_lines_view(lbl_lines* parent, int64_t first, int64_t count, lbl_lines** out);
int lbl_lines_destroy(lbl_lines* lines);
int lbl_lines_count(const lbl_lines* lines, int64_t* n);

/* ---- the hot path: Isotope.createCrossSection (pyradClasses.py:361-407) ----------- */
/* One-shot, host in / host out: per-line half-widths (pyradClasses.py:252-263), regime
 * select (pyradClasses.py:378-387), profile (pyradLineshape.py:39, 52, 58-76), intensity
 * (pyradIntensity.py:30-32), centre index (pyradClasses.py:390), accumulate
 * (pyradClasses.py:392-400) and regrid to the base grid (pyradClasses.py:401-405).
 * xsec_out has n_base elements (cm^2/molecule); regime_counts = {gaussian, lorentz, voigt}
 * as printed at pyradClasses.py:406 (may be NULL). */
int lbl_xsec_accumulate(lbl_ctx* ctx, const double* nu, const double* sw, const double* elower,
                        const double* gamma_air, const double* gamma_self, const double* n_air,
                        const double* delta_air, int64_t n_lines, const lbl_iso_params* iso,
                        const lbl_grid* grid, double* xsec_out, int64_t regime_counts[3]);

#define LBL_MAX_JOBS 65536    /* jobs (isotopologue x layer) per batched call */
/* Device-resident, asynchronous, batched form: job j accumulates lines[j] under iso[j] /
 * grid[j] into out[j] (n_base doubles).  All jobs run in one launch sequence so that
 * isotopologues, molecules and layers fill the chip together. */
int lbl_xsec_accumulate_dev(lbl_ctx* ctx, int n_jobs, lbl_lines* const* lines,
                            const lbl_iso_params* iso, const lbl_grid* grid,
                            lbl_buffer* const* out);
/* Regime counters of the most recent lbl_xsec_accumulate_dev (drains the stream):
 * counts[3*j + {0,1,2}] = {gaussian, lorentz, voigt} of job j. */
int lbl_last_regime_counts(lbl_ctx* ctx, int n_jobs, int64_t* counts);
/* Introspection for tests: the k-th most recently used dispatch schedule of the context (0 = the last one an
 * accumulate batch used).  list receives 2 ints per workgroup (job of the launch group, tile of the job), tabs 8 ints
 * per span of 64 R points {iA, iB, iC, iD, iF1, iF2, 0, 0} (the lower bounds of the span's edge / interior / far lines
 * in the job's sorted centre indices).  Either array may be NULL; *n_items / *n_tab_ints are always set.
 * built_on_device: 1 when the schedule came from the in-stream device build ("schedule_build" 1, the default),
 * 0 from the host. */
int lbl_schedule_export(lbl_ctx* ctx, int k, int32_t* list, int64_t list_cap, int32_t* tabs, int64_t tabs_cap,
                        int64_t* n_items, int64_t* n_tab_ints, int32_t* built_on_device);

/* Parity aid: runs the real line preparation (K1) of this one job and reports, per line, the centre index
 * K1 wrote (pyradClasses.py:390; the very value the accumulate kernel works from, clamped to +-2e9), and the
 * Lorentz and Doppler half-widths (pyradClasses.py:256-263), the corrected intensity (pyradIntensity.py:30-32)
 * and the regime (0 Gaussian, 1 Lorentz, 2 pseudo-Voigt; pyradClasses.py:379-387) from the same device
 * expressions K1 evaluates (a separate reporting kernel: the timed kernels carry no debug stores).
 * Any output may be NULL. */
int lbl_line_quantities(lbl_ctx* ctx, lbl_lines* lines, const lbl_iso_params* iso,
                        const lbl_grid* grid, int64_t* index, double* lorentz_hw,
                        double* gauss_hw, double* intensity, int32_t* regime);

/* ---- fused layer sweep (pyradClasses.py:566-571, 581-587, 707-716, 784-787; pyradPlanck.py:38-44)
 * For every base-grid point j:
 *   xs_m   = sum of the isotopologue cross sections of molecule m        (pyradClasses.py:566-571)
 *   k      = sum_m xs_m * conc[m] * P / 1e4 / kB / T                      (pyradClasses.py:583, 707-712)
 *   trans  = exp(-k * depth)                                             (pyradClasses.py:716)
 *   I_out  = trans * I_in + (1 - trans) * B(nu_j, T)                      (pyradClasses.py:784-787)
 * nu_j is the reference's xAxis = linspace(range_min, range_max, n, endpoint=True)
 * (pyradClasses.py:702-705).  iso_mol[i] (non-decreasing, 0-based) maps isotopologue i to
 * its molecule.  I_in == NULL with surface_T > 0 uses I_in = B(nu_j, surface_T)
 * (pyradInteractive.py:400).  Any of abs_coef / trans / I_out may be NULL.
 * Only points [first, first+count) are swept (count == 0: all n); buffers are indexed by j. */
int lbl_layer_sweep_dev(lbl_ctx* ctx, int n_iso, lbl_buffer* const* xsec, const int32_t* iso_mol,
                        int n_mol, const double* conc, double P, double T, double depth,
                        double range_min, double range_max, int64_t n,
                        int64_t first, int64_t count,
                        lbl_buffer* I_in, double surface_T,
                        lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out);

/* One step of a layer (the gas cell of pyradClasses.py:648 after its addMolecule calls):
 * lbl_xsec_accumulate_dev for the layer's n_iso line lists followed by lbl_layer_sweep_dev, in ONE
 * launch sequence.  A layer with one line list on the base grid has the sweep of a point in the
 * accumulate kernel's output stage, right after that point's cross section is final (same arithmetic,
 * bit-identical results, no sweep launch, no re-read of the cross section).  With several line lists
 * the sweep kernel follows the accumulate launch: both ways of folding it in were measured slower on
 * MI355X.  Molecule sums, absorption coefficient, transmittance, outgoing radiance:
 * pyradClasses.py:566-571, 583, 707-716, 784-787; xsec[i] always receives line list i's cross
 * sections.  All line lists share `grid` (axis, shard) and the layer's T and P (iso[i].T / .P must
 * equal iso[0]'s); iso_mol / n_mol / conc as in lbl_layer_sweep_dev; xsec[i] receives the cross
 * section of line list i.  lbl_set_option("layer_step_fused", 0) forces the two-call form. */
int lbl_layer_step_dev(lbl_ctx* ctx, int n_iso, lbl_lines* const* lines, const lbl_iso_params* iso,
                       const lbl_grid* grid, lbl_buffer* const* xsec, const int32_t* iso_mol, int n_mol,
                       const double* conc, double depth, lbl_buffer* I_in, double surface_T,
                       lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out);

/* ---- the layer's absorption coefficient accumulated directly (ABI 4) -----------------------------------------
 * Layer.absCoef is sum over molecules m of (sum over isotopologues of crossSection) * conc_m * P / 1E4 / k / T
 * (pyradClasses.py:707-712, 581-583, 566-571), and the reference computes its parts lazily (progressCrossSection,
 * pyradClasses.py:32-88).  These entry points run ONE accumulate job per LAYER: the line preparation multiplies every
 * line's amplitude by its molecule's factor f_m = conc_m P / 1E4 / k / T (host, the reference's order of operations) and
 * writes the records of all the layer's line lists into one array in centre-index order (the merged order is built on
 * the device once per (line lists, grid), beside the dispatch schedule); the accumulate kernel then sums every
 * (line, grid point) contribution of the layer into the absorption coefficient itself - one pass over the grid instead
 * of one per line list, no per-line-list cross-section arrays written and read back.  Every contribution the
 * reference's loop adds (pyradClasses.py:392-400) is still evaluated.  A per-isotopologue cross section
 * (Isotope.crossSection) is produced on demand by lbl_xsec_accumulate_dev, as before.
 * Arithmetic: fp64; differs from the per-line-list path by the order of summation and one rounding per line (the
 * factor rides on the amplitude instead of on the sum): a few 1e-16 relative.  They exist in the sweeps' default arithmetic
 * only: with "sweep_ieee_divisions" 1 lbl_layer_merged_step_dev, lbl_column_fold_dev and lbl_column_transmission answer
 * LBL_ERR_BAD_ARG (ABI 5; until then the option was silently ignored there) - use the per-line-list entry points, which honour
 * it.  "accuracy" applies as usual.  At most 64 line lists per layer (lbl_limit "merged_lists_per_job").  Needs "accum_variant" 3 or 5 and the device schedule build (LBL_ERR_BAD_ARG otherwise: use the per-list step).
 *
 * lbl_layer_merged_step_dev: one layer (gas cell) - line prep, ONE accumulate job over the merged lists, and in its
 * output stage k, transmittance exp(-k depth) and outgoing radiance trans * I_in + (1 - trans) * B(nu, T)
 * (pyradClasses.py:714-716, 784-787; pyradPlanck.py:38-44).  Arguments as lbl_layer_step_dev without the xsec buffers;
 * any of abs_coef / trans / I_out may be NULL (not all).  A work grid coarser than the base grid (dynamic resolution)
 * is regridded (np.interp of pyradClasses.py:401-405 applied to k) and swept by the sweep kernel. */
int lbl_layer_merged_step_dev(lbl_ctx* ctx, int n_iso, lbl_lines* const* lines, const lbl_iso_params* iso,
                              const lbl_grid* grid, const int32_t* iso_mol, int n_mol, const double* conc,
                              double depth, lbl_buffer* I_in, double surface_T,
                              lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out);
/* The same for a batch of layers (a column: every layer's merged job in ONE launch sequence, so that layers fill the
 * chip together): layer l owns n_iso[l] consecutive entries of lines / iso / iso_mol (molecule index 0-based inside the
 * layer, non-decreasing), n_mol[l] consecutive entries of conc, its own grid[l], and receives its absorption
 * coefficient in abs_coef[l] (n_base doubles). */
int lbl_layers_merged_accumulate_dev(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_lines* const* lines,
                                     const lbl_iso_params* iso, const lbl_grid* grid, const int32_t* iso_mol,
                                     const int32_t* n_mol, const double* conc, lbl_buffer* const* abs_coef);
/* Column step from the layers' absorption coefficients (bottom to top): per grid point and layer
 * trans = exp(-k depth), I <- trans * I + (1 - trans) * B(nu_j, T_l), I_0 = I_in or B(nu_j, surface_T)
 * (pyradClasses.py:714-716, 784-787) in one pass over the n_layers arrays.  trans: NULL, or n_layers buffers of which
 * any may be NULL (that layer's transmittance is then not written). */
int lbl_column_fold_dev(lbl_ctx* ctx, int n_layers, lbl_buffer* const* abs_coef, const double* T,
                        const double* depth, double range_min, double range_max, int64_t n,
                        int64_t first, int64_t count, lbl_buffer* I_in, double surface_T,
                        lbl_buffer* const* trans, lbl_buffer* I_out);

/* ---- resident column (ABI 5) ---------------------------------------------------------------------------------------
 * The argument blocks of a column's merged accumulate jobs (lbl_layers_merged_accumulate_dev) and of its fold
 * (lbl_column_fold_dev) kept on the C side between calls, so that a host's per-call work does not grow with layers x line
 * lists: pyrad_amd.model.Atmosphere.transmission (the fold of pyradClasses.py:784-787 over the layers' absorption
 * coefficients, pyradClasses.py:707-712) is ONE call of lbl_column_transmission and one lbl_download_wait.
 *   lbl_column_create        arguments as lbl_layers_merged_accumulate_dev plus the layers' depths; all layers share one
 *                            wavenumber range and base grid.  The handle owns nothing on the device: line lists and buffers stay
 *                            the caller's and must outlive it (or be replaced with lbl_column_set_layer first).
 *   lbl_column_set_layer     replaces one layer's line lists / parameters / grid / volume fractions / depth / buffer (what a
 *                            mutator of the reference changes: changeTemperature, changePressure, changeRange, changeDepth,
 *                            setPPM ...; the layer keeps its numbers of line lists and molecules)
 *   lbl_column_transmission  due[l] != 0 (due == NULL: every layer): layer l's absorption coefficient is recomputed (ONE merged
 *                            accumulate job per due layer, all in one launch sequence); then the fold bottom to top,
 *                            I <- T_l I + (1 - T_l) B(nu, T_l) with I_0 = I_in or B(nu, surface_T), in `pieces` pieces of the
 *                            grid, each piece's part of I_out copied to host_out (page-locked, lbl_host_alloc; may be NULL: no
 *                            download) while the next piece is folded.  Returns when everything is ENQUEUED:
 *                            lbl_download_wait(ctx) before host_out is read.  The sweeps' default arithmetic only
 *                            ("sweep_ieee_divisions" 1: LBL_ERR_BAD_ARG). */
typedef struct lbl_column lbl_column;
int lbl_column_create(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_lines* const* lines, const lbl_iso_params* iso,
                      const lbl_grid* grid, const int32_t* iso_mol, const int32_t* n_mol, const double* conc,
                      const double* depth, lbl_buffer* const* abs_coef, lbl_column** out);
int lbl_column_destroy(lbl_column* column);
int lbl_column_set_layer(lbl_column* column, int layer, lbl_lines* const* lines, const lbl_iso_params* iso,
                         const lbl_grid* grid, const double* conc, double depth, lbl_buffer* abs_coef);
int lbl_column_transmission(lbl_column* column, const uint8_t* due, lbl_buffer* I_in, double surface_T, lbl_buffer* I_out,
                            double* host_out, int pieces);

/* Column fold of Layer.transmission over layers bottom to top (pyradClasses.py:784-787):
 *   I <- trans_l * I + (1 - trans_l) * B(nu_j, layer_T[l]),  I_0 = I_in or B(nu_j, surface_T). */
int lbl_column_sweep_dev(lbl_ctx* ctx, int n_layers, lbl_buffer* const* trans, const double* layer_T,
                         double range_min, double range_max, int64_t n,
                         int64_t first, int64_t count,
                         lbl_buffer* I_in, double surface_T, lbl_buffer* I_out);

/* Column step straight from the cross sections: for every grid point, layer after layer (bottom
 * to top), the absorption coefficient and transmittance exactly as lbl_layer_sweep_dev computes
 * them, then the fold of lbl_column_sweep_dev - one pass over the cross sections instead of one
 * sweep launch per layer plus the fold.  Flattened per-layer inputs: layer l owns n_iso[l]
 * consecutive entries of xsec / iso_mol (molecule index 0-based inside the layer, non-decreasing)
 * and n_mol[l] consecutive entries of conc.  abs_coef / trans: NULL, or arrays of n_layers buffers
 * of which any entry may be NULL (that layer's array is then not written). */
int lbl_column_step_dev(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_buffer* const* xsec,
                        const int32_t* iso_mol, const int32_t* n_mol, const double* conc,
                        const double* P, const double* T, const double* depth,
                        double range_min, double range_max, int64_t n, int64_t first, int64_t count,
                        lbl_buffer* I_in, double surface_T,
                        lbl_buffer* const* abs_coef, lbl_buffer* const* trans, lbl_buffer* I_out);

/* Elementwise optical properties of a transmittance array (pyradClasses.py:73-76, 330-340,
 * 596-606, 718-732): kind 0 emissivity/emittance = 1 - T; 1 absorbance = log10(1/T);
 * 2 optical depth = -ln T. */
int lbl_optical_dev(lbl_ctx* ctx, lbl_buffer* trans, int64_t n, int kind, lbl_buffer* out);

/* out[j] = 0 + in[0][j] + in[1][j] + ... in list order: the aggregation of
 * Molecule.createCrossSection / Layer.createCrossSection (pyradClasses.py:566-571, 684-689). */
int lbl_sum_dev(lbl_ctx* ctx, int n_in, lbl_buffer* const* in, int64_t n, lbl_buffer* out);

/* Planck radiance on the layer axis (pyradPlanck.py:38-44 via pyradClasses.py:781-782). */
int lbl_planck_dev(lbl_ctx* ctx, double range_min, double range_max, int64_t n, double T,
                   lbl_buffer* out);

/* integrateSpectrum (pyradClasses.py:26-29): sum(nan_to_num(y)) * unit_angle * res.
 * Deterministic fixed-tree reduction; synchronous (returns the scalar). */
int lbl_band_integral(lbl_ctx* ctx, lbl_buffer* spectrum, int64_t n, double unit_angle, double res,
                      double* result);

/* Line survey histogram, Isotope.createLineSurvey (pyradClasses.py:409-428): raw S summed
 * into the bin of each line's centre index; out has n_base elements. */
int lbl_line_survey_dev(lbl_ctx* ctx, lbl_lines* lines, const lbl_grid* grid, lbl_buffer* out);

/* ---- graph capture ------------------------------------------------------------------ */
/* A step that repeats (same line lists, grid, buffers) can be captured once and replayed as ONE
 * hipGraph launch: the host then spends microseconds per step instead of rebuilding and checking the
 * launch sequence, which is what bounds a small shard's step on 8 GPUs.
 *   run the sequence once (allocates scratch, builds schedules, uploads descriptors);
 *   lbl_capture_begin(ctx);  the same "_dev" calls again (nothing runs, kernels are recorded);
 *   lbl_capture_end(ctx, &g);   then lbl_graph_launch(g) per step.
 * Inside a capture only kernel launches are possible: a call that would allocate, upload or
 * synchronise returns LBL_ERR_STATE (and the capture must still be ended).  The all-gather is not
 * captured (lbl_allgather_* return LBL_ERR_STATE inside a capture): enqueue it after the graph.  A graph
 * holds pointers into the context's scratch and caches and to the buffers and line lists of the captured
 * calls; lbl_graph_launch returns LBL_ERR_STATE once one of them may have changed (another batch grew a
 * scratch buffer or took a descriptor slot, or ANY buffer or line list of the context was destroyed): capture
 * again.  Page-locked host blocks (lbl_host_alloc) are never captured. */
typedef struct lbl_graph lbl_graph;
int lbl_capture_begin(lbl_ctx* ctx);
int lbl_capture_end(lbl_ctx* ctx, lbl_graph** out);
int lbl_graph_launch(lbl_graph* graph);
int lbl_graph_destroy(lbl_graph* graph);

/* ---- multi-GPU: one process per GPU, grid sharded by contiguous range --------------- */
#define LBL_UNIQUE_ID_BYTES 128
/* Rank 0 calls lbl_comm_unique_id and hands the 128 bytes to every rank out of band
 * (file, env, torch.distributed store, MPI ...); then every rank calls lbl_comm_create. */
int lbl_comm_unique_id(char id[LBL_UNIQUE_ID_BYTES]);
int lbl_comm_create(lbl_ctx* ctx, const char id[LBL_UNIQUE_ID_BYTES], int world_size, int rank,
                    lbl_comm** out);
int lbl_comm_destroy(lbl_comm* comm);
/* The single RCCL all-gather of the path: every rank contributes count doubles starting at
 * send_offset of `send`, and receives world_size*count doubles into `recv` (rank order).
 * Enqueued on the context stream (send may alias recv at its own slot: in-place). */
int lbl_allgather_dev(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count,
                      lbl_buffer* recv);
/* Same collective, but the context stream does not wait for it: kernels enqueued afterwards
 * (the next step, on OTHER buffers) overlap the transfer.  `slot` (0..6) names the completion
 * event; lbl_comm_fence_dev(comm, slot) makes the context stream wait (no host sync) for the
 * collective issued with that slot, slot -1 for all of them: call it before send/recv of that
 * collective are touched again.  Collectives run, in issue order, on a stream the communicator owns.
 * "The context stream" is the stream of the context that owns send/recv (both the same, any context of
 * the communicator's device): several contexts of one process - independent steps in flight on streams
 * of their own - share ONE communicator, whose order of collectives stays the same on every rank.  A
 * slot belongs to the context that used it last; reusing it from another context before its fence is
 * LBL_ERR_STATE. */
int lbl_allgather_overlap_dev(lbl_comm* comm, lbl_buffer* send, int64_t send_offset, int64_t count,
                              lbl_buffer* recv, int slot);
int lbl_comm_fence_dev(lbl_comm* comm, int slot);
/* Fewer, larger collectives: stage the shards of several consecutive steps side by side in one batch
 * buffer (rank r, step b of B at [(r*B + b)*S, +S)) and send them with ONE all-gather of B*S doubles per
 * rank.  This is the staging copy: dst[dst_offset .. +n) = src[src_offset .. +n), device to device, async
 * on the stream of the context that owns dst. */
int lbl_gather_stage_dev(lbl_buffer* dst, int64_t dst_offset, lbl_buffer* src, int64_t src_offset, int64_t n);
/* Cost-balanced (unequal) shards: every rank sends `slot` doubles starting at its own first point,
 * so the gathered buffer holds rank r's shard in the first count[r] entries of slot r.  This puts
 * it back in grid order: out[first[r] + i] = gathered[r * slot + i], i < count[r], r < world_size
 * (first / count: host arrays of world_size entries; out at least max(first[r] + count[r]) long). */
int lbl_gather_compact_dev(lbl_ctx* ctx, lbl_buffer* gathered, int world_size, int64_t slot,
                           const int64_t* first, const int64_t* count, lbl_buffer* out);

#ifdef __cplusplus
}
#endif
#endif /* PYRAD_HIP_H */
