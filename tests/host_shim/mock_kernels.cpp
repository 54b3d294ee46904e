// CPU stand-ins for the launchers of pyrad_amd/csrc/lbl_kernels.hip (declared in lbl_device.h), for the sanitizer harness of the
// host shim.  TEST INFRASTRUCTURE, never linked into libpyrad_hip.so.  No physics here: every stand-in READS and WRITES exactly
// the index ranges its kernel reads and writes (quoted beside each one), so that AddressSanitizer - "device" memory is host
// memory in this build - catches an arena, buffer, descriptor block, span table or dispatch list that the shim sized or
// indexed wrongly; consistency of the descriptors themselves is asserted (abort on violation).
#include "../../include/pyrad_hip.h"
#include "lbl_device.h"
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace mockhip {
int device_count = 1;
long long fail_malloc_at = -1;
long long clock = 0;
long long live_allocs = 0, live_streams = 0, live_events = 0, live_graphs = 0, live_host = 0;
}

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "mock kernel: violated: %s (%s:%d)\n", #cond, __FILE__, __LINE__); abort(); } } while (0)

namespace lbl {

static volatile double g_sink;
static void touch(const double* p, long long first, long long count) { if (count > 0) g_sink = p[first] + p[first + count - 1]; }

void comm_quiesce(lbl_ctx*) {}
void comm_forget(lbl_ctx*) {}

static long long centre_index(double nu, double range_min, double resolution) {
    double q = (nu - range_min) / resolution;
    if (!(q > -2e9)) q = -2e9;
    if (!(q < 2e9)) q = 2e9;
    return (long long)q;
}

static void prep_one(const PrepJob& J, int i, HotRec& r, ColdRec& c, int32_t& ci) {
    // reads line i of all seven fields
    g_sink = J.nu[i] + J.sw[i] + J.elower[i] + J.gamma_air[i] + J.gamma_self[i] + J.n_air[i] + J.delta_air[i];
    const long long idx = centre_index(J.nu[i], J.range_min, J.resolution);
    r.cf = (double)idx; r.a2 = 1.0; r.KL = J.sw[i] * J.weight; r.dgi = 0; r.flags = 0;
    c.KG = 0.0; c.b = 1.0; c.q2 = -1.0; c.KLd = r.KL;
    ci = (int32_t)idx;
}

// line_prep_kernel: grid (ceil(max_lines / 256), n_jobs): job j writes hot / cold / cidx [0, n_lines) and block_counts[bx * 3 + k]
// for EVERY block bx of the grid (blocks beyond its own lines store zeros); a list of a merged job is skipped
void launch_line_prep(const PrepJob* d_jobs, int n_jobs, int max_lines, hipStream_t) {
    if (n_jobs <= 0 || max_lines <= 0) return;
    // (self-test of the harness, tests/test_host_shim_asan_cpu.py: with SHIM_INJECT_OVERRUN set the stand-in writes one block
    // of counters more than the kernel does - AddressSanitizer must report it)
    static const int extra = getenv("SHIM_INJECT_OVERRUN") ? 4096 : 0;
    const int blocks = (max_lines + 255) / 256 + extra;
    for (int j = 0; j < n_jobs; ++j) {
        const PrepJob& J = d_jobs[j];
        if (J.merged) continue;
        CHECK(J.n_lines <= max_lines);
        for (int i = 0; i < J.n_lines; ++i) prep_one(J, i, J.hot[i], J.cold[i], J.cidx[i]);
        for (int bx = 0; bx < blocks; ++bx)
            for (int k = 0; k < 3; ++k) J.block_counts[bx * 3 + k] = (k == 1) ? (unsigned)std::max(0, std::min(256, J.n_lines - bx * 256)) : 0u;
    }
}

// line_prep_merged_kernel: job m writes its record arrays [0, n_total) from src[0, n_total) and, for its blocks [0, blocks),
// block_counts[bx * 3 + k] of each of its lists
void launch_line_prep_merged(const PrepJob* d_lists, const MergedPrep* d_jobs, int n_jobs, int max_total, hipStream_t) {
    if (n_jobs <= 0 || max_total <= 0) return;
    for (int m = 0; m < n_jobs; ++m) {
        const MergedPrep& M = d_jobs[m];
        CHECK(M.n_total <= max_total && M.blocks == (M.n_total + 255) / 256 && M.n_lists >= 1 && M.n_lists <= kMaxIso);
        std::vector<std::vector<unsigned>> cnt((size_t)M.n_lists, std::vector<unsigned>((size_t)M.blocks, 0u));
        for (int t = 0; t < M.n_total; ++t) {
            const int s = M.src[t];
            const int list = (int)((unsigned)s >> 26), line = s & ((1 << 26) - 1);
            CHECK(list < M.n_lists);
            const PrepJob& J = d_lists[M.first_list + list];
            CHECK(J.merged && line < J.n_lines);
            prep_one(J, line, M.hot[t], M.cold[t], M.cidx[t]);
            cnt[(size_t)list][(size_t)(t / 256)]++;
        }
        for (int l = 0; l < M.n_lists; ++l)
            for (int bx = 0; bx < M.blocks; ++bx)
                for (int k = 0; k < 3; ++k) d_lists[M.first_list + l].block_counts[bx * 3 + k] = k == 1 ? cnt[(size_t)l][(size_t)bx] : 0u;
    }
}

void launch_line_quantities(const PrepJob* d_job, int n_lines, long long* index, double* lhw, double* ghw, double* intensity,
                            int32_t* regime, hipStream_t) {
    for (int i = 0; i < n_lines; ++i) { index[i] = d_job->cidx[i]; lhw[i] = 0.07; ghw[i] = 7e-4; intensity[i] = d_job->sw[i]; regime[i] = 1; }
}

// centre_index_kernel + merge_rank_kernel: every list writes tmp_cidx[0, n_lines); every line of a job writes one entry of
// src_of_job[0, total lines of the job): the inverse of the stable merge, ties by list
void launch_merge_ranks(const MergeList* d_lists, int n_lists, int max_lines, hipStream_t) {
    for (int a = 0; a < n_lists; ++a) {
        const MergeList& A = d_lists[a];
        CHECK(A.n_lines <= max_lines && A.job_first >= 0 && A.job_first + A.job_count <= n_lists && a >= A.job_first && a < A.job_first + A.job_count);
        for (int i = 0; i < A.n_lines; ++i) A.tmp_cidx[i] = (int32_t)centre_index(A.nu[i], A.range_min, A.resolution);
    }
    for (int a = 0; a < n_lists; ++a) {
        const MergeList& A = d_lists[a];
        for (int i = 0; i < A.n_lines; ++i) {
            const int c = A.tmp_cidx[i];
            long long pos = i;
            for (int b = A.job_first; b < A.job_first + A.job_count; ++b) {
                if (b == a) continue;
                const MergeList& B = d_lists[b];
                const int32_t* lo = B.tmp_cidx, *hi = B.tmp_cidx + B.n_lines;
                pos += (b < a) ? (std::upper_bound(lo, hi, c) - lo) : (std::lower_bound(lo, hi, c) - lo);
            }
            A.src_of_job[pos] = (int32_t)(((unsigned)(a - A.job_first) << 26) | (unsigned)i);
        }
    }
}

// sched_spans_kernel + the order kernels: tabs[8 * total_spans] (rows of the jobs' span tables, job-major), scratch as laid
// out by sched_scratch_bytes, worklist[sched_launch_items(total_tiles, n_cu, xcd_pack)] (fillers hold (0, -1))
void launch_schedule_build(const SchedJob* d_jobs, int n_jobs, int total_spans, int total_tiles, int R, int spans_per_tile,
                           long long, double, double, double, double, int n_cu, int32_t* tabs, void* scratch, int2* worklist,
                           hipStream_t, int, bool xcd_pack, int, int, int) {
    if (total_spans <= 0 || total_tiles <= 0) return;
    CHECK(sched_device_supported(total_tiles, n_cu));
    memset(scratch, 0, sched_scratch_bytes(total_tiles));
    int spans = 0, tiles = 0;
    const int items = sched_launch_items(total_tiles, n_cu, xcd_pack);
    for (int i = 0; i < items; ++i) worklist[i] = int2{0, -1};
    for (int j = 0; j < n_jobs; ++j) {
        const SchedJob& J = d_jobs[j];
        CHECK(J.span_first == spans && J.tile_first == tiles && J.p_end >= J.p_begin);
        const long long pts = (long long)J.p_end - J.p_begin;
        const int ns = (int)((pts + 64LL * R - 1) / (64LL * R)), nt = (ns + spans_per_tile - 1) / spans_per_tile;
        for (int i = 0; i < J.n_lines; i += std::max(1, J.n_lines / 7)) g_sink = J.cidx[i];
        if (J.n_lines > 0) g_sink = J.cidx[J.n_lines - 1];
        for (int q = 0; q < ns; ++q) {
            int32_t* row = tabs + (size_t)(spans + q) * 8;
            row[0] = 0; row[1] = 0; row[2] = J.n_lines; row[3] = J.n_lines; row[4] = 0; row[5] = J.n_lines; row[6] = 0; row[7] = J.n_lines;
        }
        for (int t = 0; t < nt; ++t) worklist[tiles + t] = int2{j, t};
        spans += ns; tiles += nt;
    }
    CHECK(spans == total_spans && tiles == total_tiles);
}

// the accumulate kernels: job j reads its records [0, n_lines), its span-table rows, writes out[p_begin, p_end) (if any) and the
// fused sweep's arrays at the same points; the dispatch list has total_tiles entries (job, tile), tile -1: an idle position
static void accumulate_common(const AccumJob* d_jobs, int n_jobs, int max_tiles, int tile_points, int span_points, const int2* worklist,
                              int total_tiles) {
    if (worklist) {
        for (int i = 0; i < total_tiles; ++i) {
            const int2 w = worklist[i];
            CHECK(w.x >= 0 && w.x < n_jobs && w.y >= -1 && w.y < std::max(d_jobs[w.x].n_tiles, 1));
        }
    }
    for (int j = 0; j < n_jobs; ++j) {
        const AccumJob& J = d_jobs[j];
        CHECK(J.p_begin >= 0 && J.p_end >= J.p_begin && J.p_end <= J.n_work && J.n_tiles <= max_tiles);
        CHECK((long long)J.n_tiles * tile_points >= (long long)J.p_end - J.p_begin);
        if (J.n_lines > 0) {
            g_sink = J.hot[0].cf + J.hot[J.n_lines - 1].cf + J.cold[0].KG + J.cold[J.n_lines - 1].KG + J.cidx[0] + J.cidx[J.n_lines - 1];
            for (int i = 1; i < J.n_lines; ++i) CHECK(J.cidx[i] >= J.cidx[i - 1]);      // the kernels rely on sorted centre indices
        }
        if (J.span_tab) {
            const long long ns = ((long long)J.p_end - J.p_begin + span_points - 1) / span_points;
            for (long long q = 0; q < ns; ++q) {
                const int32_t* row = J.span_tab + (size_t)q * 8;
                CHECK(row[0] >= 0 && row[0] <= row[1] && row[1] <= row[2] && row[2] <= row[3] && row[3] <= J.n_lines);
                // interior lines: far-left | Lorentz term from the series, Gaussian part in the near walk | near | ... | far-right
                CHECK(row[1] <= row[4] && row[4] <= row[6] && row[6] <= row[7] && row[7] <= row[5] && row[5] <= row[2]);
            }
        }
        for (long long p = J.p_begin; p < J.p_end; ++p) {
            if (J.out) J.out[p] = 0.0;
            if (J.fuse.on) {
                if (J.fuse.I_in) g_sink = J.fuse.I_in[p];
                if (J.fuse.abs_coef) J.fuse.abs_coef[p] = 0.0;
                if (J.fuse.trans) J.fuse.trans[p] = 1.0;
                if (J.fuse.I_out) J.fuse.I_out[p] = 0.0;
                CHECK(p < J.fuse.n);
            }
        }
    }
}

void launch_accumulate(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, int LS, int variant, const int2* worklist,
                       int total_tiles, hipStream_t, int, int gauss_run) {
    if (n_jobs <= 0 || max_tiles <= 0) return;
    CHECK((R == 1 || R == 2 || R == 4 || R == 8) && (LS == 1 || LS == 2 || LS == 4 || LS == 8) && (gauss_run == 16 || gauss_run == 32));
    CHECK(gauss_run == 16 || (variant >= 5 && R == 4 && LS == 1));
    accumulate_common(d_jobs, n_jobs, max_tiles, accumulate_tile_points(R, LS, variant), 64 * R, variant >= 3 ? worklist : nullptr, total_tiles);
}

void launch_accumulate_skew(const AccumJob* d_jobs, int n_jobs, int max_tiles, int R, const int2* worklist, int total_tiles,
                            hipStream_t, int LS) {
    if (n_jobs <= 0 || max_tiles <= 0) return;
    CHECK((R == 1 || R == 2 || R == 4 || R == 8) && (LS == 1 || (R == 8 && (LS == 2 || LS == 4))));
    accumulate_common(d_jobs, n_jobs, max_tiles, accumulate_tile_points(R, LS, 3), 64 * R, worklist, total_tiles);
}

int balanced_workers(int, int) { return 0; }
void launch_accumulate_balanced(const AccumJob*, int, int, int, int, SpanRec*, unsigned int*, unsigned long long*, double*, hipStream_t) {}

void accumulate_far_field_params(int R, int* far_half_spans, double* far_cost, int) {
    *far_half_spans = 4;
    *far_cost = (3.0 * 17.0 + 12.0) / 64.0 / (5.0 * R);
}

// regrid_kernel: reads work[0, n_work), writes out[0, n_base)
void launch_regrid(const double* work, long long n_work, double* out, long long n_base, double, double, hipStream_t) {
    if (n_base <= 0) return;
    touch(work, 0, n_work);
    for (long long j = 0; j < n_base; ++j) out[j] = 0.0;
}

// layer_sweep_kernel: reads xsec[i][first, first + count) of its n_iso terms, writes abs_coef / trans / I_out there
void launch_layer_sweep(const SweepArgs& a, hipStream_t) {
    CHECK(a.n_iso >= 0 && a.n_iso <= kMaxIso && a.first >= 0 && a.count >= 0 && a.first + a.count <= a.n);
    for (int i = 0; i < a.n_iso; ++i) touch(a.xsec[i], a.first, a.count);
    if (a.I_in) touch(a.I_in, a.first, a.count);
    for (long long j = a.first; j < a.first + a.count; ++j) {
        if (a.abs_coef) a.abs_coef[j] = 0.0;
        if (a.trans) a.trans[j] = 1.0;
        if (a.I_out) a.I_out[j] = 0.0;
    }
}

// column_step_kernel: reads every term's array over [first, first + count), writes I_out there and the layers' optional arrays
void launch_column_step(const ColumnStepArgs* d_args, long long first, long long count, hipStream_t, int, int) {
    if (count <= 0) return;
    const ColumnStepArgs& A = *d_args;
    CHECK(A.n_terms >= 0 && A.n_terms <= kMaxColumnIso && A.n_layers >= 0 && A.n_layers <= kMaxLayers && first >= 0 && first + count <= A.n);
    int layers = 0;
    for (int t = 0; t < A.n_terms; ++t) {
        touch(A.xsec[t], first, count);
        if (A.term_flags[t] & TERM_LAST_LAYER) {
            CHECK(A.term_flags[t] & TERM_LAST_MOL);
            if (A.layer_arrays) {
                CHECK(layers < kMaxLayers);
                for (long long j = first; j < first + count; ++j) {
                    if (A.abs_coef[layers]) A.abs_coef[layers][j] = 0.0;
                    if (A.trans[layers]) A.trans[layers][j] = 1.0;
                }
            }
            ++layers;
        }
    }
    CHECK(A.n_terms == 0 || layers == A.n_layers);
    CHECK(A.pbkT_min <= A.pbkT_max);
    if (A.I_in) touch(A.I_in, first, count);
    for (long long j = first; j < first + count; ++j) A.I_out[j] = 0.0;
}

void launch_column_sweep(const ColumnArgs* d_args, long long count, hipStream_t, int) {
    if (count <= 0) return;
    const ColumnArgs& A = *d_args;
    CHECK(A.n_layers >= 0 && A.n_layers <= kMaxLayers && A.first >= 0 && A.count == count && A.first + A.count <= A.n);
    for (int l = 0; l < A.n_layers; ++l) touch(A.trans[l], A.first, A.count);
    if (A.I_in) touch(A.I_in, A.first, A.count);
    for (long long j = A.first; j < A.first + A.count; ++j) A.I_out[j] = 0.0;
}

void launch_planck(double* out, long long n, double, double, double, double, double, double, hipStream_t) { for (long long j = 0; j < n; ++j) out[j] = 1.0; }

// band_partial_kernel + band_final_kernel: partial[band_partial_count(n)], result[1]
void launch_band_integral(const double* y, long long n, double* partial, double* result, hipStream_t) {
    touch(y, 0, n);
    for (int b = 0; b < band_partial_count(n); ++b) partial[b] = 0.0;
    result[0] = 0.0;
}

void launch_sum(const SumArgs& a, hipStream_t) {
    CHECK(a.n_in >= 0 && a.n_in <= kMaxIso);
    for (int i = 0; i < a.n_in; ++i) touch(a.in[i], 0, a.n);
    for (long long j = 0; j < a.n; ++j) a.out[j] = 0.0;
}

// gather_compact_kernel: rank r's `count[r]` doubles from gathered[r * slot ...) to out[first[r] ...)
void launch_gather_compact(const CompactArgs& a, long long max_count, hipStream_t) {
    CHECK(a.world >= 1 && a.world <= kMaxRanks);
    for (int r = 0; r < a.world; ++r) {
        CHECK(a.count[r] <= max_count && a.count[r] <= a.slot);
        for (long long i = 0; i < a.count[r]; ++i) a.out[a.first[r] + i] = a.gathered[(long long)r * a.slot + i];
    }
}

void launch_optical(const double* trans, long long n, int kind, double* out, hipStream_t) {
    CHECK(kind >= 0 && kind <= 2);
    for (long long j = 0; j < n; ++j) out[j] = trans[j];
}

// line_survey_kernel: reads nu / sw [0, n_lines), adds into out[0, n_base)
void launch_line_survey(const double* nu, const double* sw, int n_lines, double range_min, double resolution, double* out,
                        long long n_base, hipStream_t) {
    for (int i = 0; i < n_lines; ++i) {
        const long long c = centre_index(nu[i], range_min, resolution);
        if (c >= 0 && c < n_base) out[c] += sw[i];
    }
}

}  // namespace lbl
