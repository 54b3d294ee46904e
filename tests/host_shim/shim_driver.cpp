// Sanitizer driver of the host shim: pyrad_amd/csrc/lbl_api.hip (contexts, pools, descriptor caches, the schedule cache and its
// host-side builder, line-list views, launch-shape choices, argument checks, the resident column) compiled as plain C++ with
// -fsanitize=address,undefined against the stand-in HIP runtime (mock/hip/hip_runtime.h) and launchers that touch exactly the
// ranges the kernels touch (mock_kernels.cpp), driven through the PUBLIC C ABI with seeded random cells, columns, shards, option
// settings, bad arguments, injected allocation failures and destruction orders.  TEST INFRASTRUCTURE (SURVEY.md §5 "race
// detection / sanitizers"; round-5 verdict, item 7).  usage: shim_driver <seed> <rounds>
#include "../../include/pyrad_hip.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static std::mt19937_64 rng;
static long long n_calls = 0, n_refused = 0;
static int urand(int lo, int hi) { return lo + (int)(rng() % (unsigned long long)(hi - lo + 1)); }
static double frand(double lo, double hi) { return lo + (hi - lo) * (double)(rng() >> 11) / 9007199254740992.0; }
static bool coin(double p = 0.5) { return frand(0, 1) < p; }

#define MUST(call) do { int rc_ = (call); ++n_calls; if (rc_ != LBL_OK) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, lbl_last_error(ctx)); exit(2); } } while (0)
// a call that may be refused (bad argument on purpose, injected allocation failure): any status is fine, a crash or a sanitizer report is not
static long long g_site_ok[1024], g_site_no[1024];
#define MAY(call) do { int rc_ = (call); ++n_calls; if (rc_ != LBL_OK) { ++n_refused; ++g_site_no[__LINE__ % 1024]; if (getenv("SHIM_VERBOSE")) fprintf(stderr, "line %d refused %d: %s\n", __LINE__, rc_, lbl_last_error(ctx)); } else ++g_site_ok[__LINE__ % 1024]; } while (0)
#define REFUSED(call) do { int rc_ = (call); ++n_calls; ++n_refused; if (rc_ == LBL_OK) { fprintf(stderr, "%s:%d: %s was accepted\n", __FILE__, __LINE__, #call); exit(3); } } while (0)

struct Cell {
    lbl_grid grid;
    double T, P;
};

static Cell random_cell() {
    Cell c;
    const double res_opts[] = {0.01, 0.001, 0.1};
    const double base = coin(0.7) ? 0.01 : 0.001;
    double res = coin(0.75) ? base : res_opts[urand(0, 2)];
    if (res < base) res = base;
    c.P = coin(0.3) ? 1013.25 : frand(2.0, 1500.0);
    c.T = (double)urand(200, 320);
    const double lo = (double)urand(0, 2000), width = coin(0.2) ? frand(0.5, 3.0) : frand(5.0, 60.0);
    const double hi = lo + width;
    const double dfc = 5.0 * c.P / 1013.25;
    lbl_grid g;
    g.range_min = lo; g.range_max = hi; g.resolution = res; g.base_resolution = base;
    g.n_work = (int64_t)((hi - lo) / res); g.n_base = (int64_t)((hi - lo) / base);
    g.window = (int64_t)std::ceil(dfc / res);
    g.shard_first = 0; g.shard_count = 0;
    if (res == base && g.n_work > 64 && coin(0.3)) {        // a contiguous shard of the grid, as a rank of a sharded run computes
        g.shard_first = urand(0, (int)g.n_work / 2);
        g.shard_count = urand(1, (int)(g.n_work - g.shard_first));
    }
    c.grid = g;
    return c;
}

struct Lines { lbl_lines* h; int64_t n; double lo, hi; };

static Lines make_lines(lbl_ctx* ctx, int n, double lo, double hi) {
    std::vector<double> f[7];
    for (auto& v : f) v.resize((size_t)std::max(n, 1));
    for (int i = 0; i < n; ++i) f[0][(size_t)i] = frand(lo, hi);
    std::sort(f[0].begin(), f[0].begin() + n);
    if (n > 3 && coin(0.3)) f[0][1] = f[0][0];              // duplicated wavenumbers
    for (int i = 0; i < n; ++i) {
        f[1][(size_t)i] = std::pow(10.0, frand(-28, -19)); f[2][(size_t)i] = frand(-1, 5000); f[3][(size_t)i] = coin(0.05) ? 0.0 : frand(0.05, 0.1);
        f[4][(size_t)i] = coin(0.05) ? 0.0 : frand(0.06, 0.12); f[5][(size_t)i] = frand(-0.3, 0.8); f[6][(size_t)i] = frand(-0.01, 0.005);
    }
    Lines L{nullptr, n, lo, hi};
    MUST(lbl_lines_create(ctx, f[0].data(), f[1].data(), f[2].data(), f[3].data(), f[4].data(), f[5].data(), f[6].data(), n, &L.h));
    return L;
}

static const struct { const char* key; std::vector<int> good; int bad; } kOptions[] = {
    {"accum_variant", {0, 3, 5}, 7}, {"accum_points_per_lane", {0, 1, 2, 4, 8}, 3}, {"accum_longest_first", {0, 1, 2, 3, 4}, 9},
    {"accum_tile_order", {0, 1, 2}, 5}, {"accum_line_split", {0, 1, 2, 4, 8}, 3}, {"accum_skew", {0, 1, 2}, 4},
    {"accum_skew_points_per_lane", {1, 2, 4, 8}, 3}, {"accum_xcd_chunks", {0, 1, 10, 32, 64}, 99}, {"accum_xcd_tolerance", {-1, 0, 3, 15}, 40},
    {"accum_xcd_pack", {0, 1, 2, 3}, 8}, {"accum_skew_line_split", {0, 1, 2, 4}, 3}, {"accum_far_min_window", {0, 500, 2000}, -4},
    {"accum_gauss_run", {0, 16, 32}, 8}, {"accuracy", {0, 1}, 2}, {"sweep_ieee_divisions", {0, 1}, 2}, {"schedule_build", {0, 1}, 2},
    {"layer_step_fused", {0, 1}, 2},
};

static void random_options(lbl_ctx* ctx, double p) {
    for (const auto& o : kOptions) {
        if (!coin(p)) continue;
        MUST(lbl_set_option(ctx, o.key, o.good[(size_t)urand(0, (int)o.good.size() - 1)]));
        if (coin(0.1)) REFUSED(lbl_set_option(ctx, o.key, o.bad));
    }
    // (the all-direct scalar kernels and positional orders are slow paths on a GPU, not here: everything is fair game)
    if (coin(0.05)) REFUSED(lbl_set_option(ctx, "no_such_option", 1));
}

static void default_options(lbl_ctx* ctx) {
    const char* keys[] = {"accum_points_per_lane", "accum_line_split", "accum_xcd_chunks", "accum_far_min_window", "accum_gauss_run", "accum_skew_line_split",
                          "accuracy", "sweep_ieee_divisions"};
    for (const char* k : keys) MUST(lbl_set_option(ctx, k, 0));
    MUST(lbl_set_option(ctx, "accum_variant", 5)); MUST(lbl_set_option(ctx, "accum_longest_first", 4)); MUST(lbl_set_option(ctx, "accum_tile_order", 1));
    MUST(lbl_set_option(ctx, "accum_skew", 1)); MUST(lbl_set_option(ctx, "accum_skew_points_per_lane", 8)); MUST(lbl_set_option(ctx, "accum_xcd_tolerance", 3));
    MUST(lbl_set_option(ctx, "accum_xcd_pack", 1)); MUST(lbl_set_option(ctx, "schedule_build", 1)); MUST(lbl_set_option(ctx, "layer_step_fused", 1));
}

static void one_round(int round) {
    lbl_ctx* ctx = nullptr;
    MUST(lbl_ctx_create(0, &ctx));
    { lbl_ctx* none = nullptr; int rc = lbl_ctx_create(5, &none); ++n_calls; ++n_refused; if (rc == LBL_OK) exit(3); }
    char name[64]; int n_cu = 0; int64_t hbm = 0;
    MUST(lbl_device_info(ctx, name, sizeof name, &n_cu, &hbm));
    if (coin(0.5)) random_options(ctx, 0.3);
    if (coin(0.3)) MUST(lbl_profile_enable(ctx, 1));
    std::vector<lbl_buffer*> bufs;
    std::vector<Lines> lists, views;
    auto buffer = [&](int64_t n) { lbl_buffer* b = nullptr; MUST(lbl_buffer_create(ctx, n, &b)); bufs.push_back(b); return b; };

    const int n_cells = urand(1, 4);
    for (int cell = 0; cell < n_cells; ++cell) {
        Cell c = random_cell();
        const lbl_grid& g = c.grid;
        const int64_t n = g.n_base;
        const double dfc = 5.0 * c.P / 1013.25;
        // line lists over the window, and windows of them as views
        const int n_lists = urand(1, coin(0.15) ? 70 : 5);
        std::vector<lbl_lines*> L;
        std::vector<lbl_iso_params> iso;
        std::vector<int32_t> iso_mol;
        std::vector<double> conc;
        int mol = 0;
        for (int i = 0; i < n_lists; ++i) {
            const int nl = coin(0.1) ? 0 : urand(1, coin(0.2) ? 6000 : 400);
            Lines base = make_lines(ctx, nl, std::max(g.range_min - dfc, 0.0), g.range_max + dfc);
            lists.push_back(base);
            lbl_lines* use = base.h;
            if (nl > 4 && coin(0.4)) {
                const int64_t first = urand(0, nl / 2), count = urand(0, (int)(nl - first));
                Lines v{nullptr, count, 0, 0};
                MUST(lbl_lines_view(base.h, first, count, &v.h));
                views.push_back(v);
                use = v.h;
                lbl_lines* bad = nullptr;
                if (coin(0.2)) REFUSED(lbl_lines_view(base.h, first, (int64_t)nl + 1, &bad));
                if (coin(0.2)) REFUSED(lbl_lines_view(base.h, -1, 1, &bad));
                if (coin(0.1)) REFUSED(lbl_lines_view(base.h, INT64_MAX - 2, 8, &bad));
                if (coin(0.1)) REFUSED(lbl_lines_destroy(base.h));              // a list with a live view is not destroyed
            }
            L.push_back(use);
            if (i > 0 && coin(0.5)) ++mol;
            iso_mol.push_back(mol);
            if ((int)conc.size() <= mol) conc.push_back(frand(1e-6, 1e-2));
            iso.push_back(lbl_iso_params{c.T, c.P, conc[(size_t)mol], frand(16, 48), frand(100, 3000), frand(100, 3000)});
        }
        const int n_mol = mol + 1;
        std::vector<lbl_buffer*> xs;
        for (int i = 0; i < n_lists; ++i) xs.push_back(buffer(n));
        lbl_buffer* k = buffer(n), *tr = buffer(n), *I = buffer(n), *Iin = buffer(n);
        MUST(lbl_buffer_fill(Iin, 1.0));
        std::vector<lbl_grid> grids((size_t)n_lists, g);
        // per-line-list accumulate, twice (the second call finds schedules and descriptors cached), with other options between
        MAY(lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data()));
        if (coin(0.5)) random_options(ctx, 0.15);
        MAY(lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data()));
        std::vector<int64_t> counts((size_t)n_lists * 3);
        MAY(lbl_last_regime_counts(ctx, n_lists, counts.data()));
        {   // what the tests read back: the schedule's dispatch list and span table
            int64_t ni = 0, nt = 0; int32_t dev = 0;
            if (lbl_schedule_export(ctx, 0, nullptr, 0, nullptr, 0, &ni, &nt, &dev) == LBL_OK && ni > 0) {
                std::vector<int32_t> list((size_t)ni * 2), tabs((size_t)std::max<int64_t>(nt, 1));
                MAY(lbl_schedule_export(ctx, 0, list.data(), ni * 2, tabs.data(), nt, &ni, &nt, &dev));
                MAY(lbl_schedule_export(ctx, 0, list.data(), 1, tabs.data(), 1, &ni, &nt, &dev));
            }
            ++n_calls;
        }
        // the layer steps.  A sharded grid sweeps its own range only.
        MAY(lbl_layer_step_dev(ctx, n_lists, L.data(), iso.data(), &g, xs.data(), iso_mol.data(), n_mol, conc.data(), 10.0, coin() ? Iin : nullptr, 288.0, k, tr, I));
        MAY(lbl_layer_merged_step_dev(ctx, n_lists, L.data(), iso.data(), &g, iso_mol.data(), n_mol, conc.data(), 10.0, nullptr, 288.0, k, coin() ? tr : nullptr, coin() ? I : nullptr));
        MAY(lbl_layer_sweep_dev(ctx, n_lists, xs.data(), iso_mol.data(), n_mol, conc.data(), c.P, c.T, 10.0, g.range_min, g.range_max, n, 0, 0, nullptr, 288.0, k, tr, coin() ? I : nullptr));
        if (n > 8) MAY(lbl_layer_sweep_dev(ctx, n_lists, xs.data(), iso_mol.data(), n_mol, conc.data(), c.P, c.T, 10.0, g.range_min, g.range_max, n, 3, n - 7, Iin, 0.0, k, tr, I));
        REFUSED(lbl_layer_sweep_dev(ctx, n_lists, xs.data(), iso_mol.data(), n_mol, conc.data(), c.P, c.T, 10.0, g.range_min, g.range_max, n, 1, n, Iin, 0.0, k, tr, I));
        REFUSED(lbl_layer_sweep_dev(ctx, n_lists, xs.data(), iso_mol.data(), n_mol, conc.data(), c.P, -1.0, 10.0, g.range_min, g.range_max, n, 0, 0, Iin, 0.0, k, tr, I));
        if (n > 0) { lbl_buffer* shorty = buffer(n - 1); REFUSED(lbl_layer_sweep_dev(ctx, n_lists, xs.data(), iso_mol.data(), n_mol, conc.data(), c.P, c.T, 10.0, g.range_min, g.range_max, n, 0, 0, Iin, 0.0, shorty, tr, I)); }
        // a column of this cell's line lists at other pressures: batched merged jobs + the fold, the column handle, the per-list column step
        if (g.shard_count == 0 && n_lists <= 64 && coin(0.7)) {
            const int nl = urand(1, 5);
            std::vector<int32_t> c_niso, c_nmol, c_isomol;
            std::vector<lbl_lines*> c_lines;
            std::vector<lbl_iso_params> c_iso;
            std::vector<lbl_grid> c_grid;
            std::vector<double> c_conc, c_T, c_depth;
            std::vector<lbl_buffer*> c_k, c_tr;
            for (int l = 0; l < nl; ++l) {
                const double P = c.P * std::pow(0.6, l);
                lbl_grid gl = g;
                gl.window = (int64_t)std::ceil(5.0 * P / 1013.25 / gl.resolution);
                c_grid.push_back(gl); c_niso.push_back(n_lists); c_nmol.push_back(n_mol);
                for (int i = 0; i < n_lists; ++i) { lbl_iso_params p = iso[(size_t)i]; p.P = P; p.T = c.T - 5 * l; c_iso.push_back(p); c_lines.push_back(L[(size_t)i]); c_isomol.push_back(iso_mol[(size_t)i]); }
                c_conc.insert(c_conc.end(), conc.begin(), conc.end());
                c_T.push_back(c.T - 5 * l); c_depth.push_back(100.0 * (l + 1));
                c_k.push_back(buffer(n)); c_tr.push_back(coin() ? buffer(n) : nullptr);
            }
            MAY(lbl_layers_merged_accumulate_dev(ctx, nl, c_niso.data(), c_lines.data(), c_iso.data(), c_grid.data(), c_isomol.data(), c_nmol.data(), c_conc.data(), c_k.data()));
            MAY(lbl_column_fold_dev(ctx, nl, c_k.data(), c_T.data(), c_depth.data(), g.range_min, g.range_max, n, 0, 0, nullptr, 288.0, c_tr.data(), I));
            if (n > 16) MAY(lbl_column_fold_dev(ctx, nl, c_k.data(), c_T.data(), c_depth.data(), g.range_min, g.range_max, n, 4, n - 9, Iin, 0.0, nullptr, I));
            lbl_column* col = nullptr;
            if (lbl_column_create(ctx, nl, c_niso.data(), c_lines.data(), c_iso.data(), c_grid.data(), c_isomol.data(), c_nmol.data(), c_conc.data(), c_depth.data(), c_k.data(), &col) == LBL_OK) {
                void* host = nullptr;
                MUST(lbl_host_alloc(ctx, std::max<int64_t>(n, 1) * 8, &host));
                std::vector<uint8_t> due((size_t)nl);
                for (auto& d : due) d = coin() ? 1 : 0;
                MAY(lbl_column_transmission(col, nullptr, nullptr, 288.0, I, (double*)host, urand(1, 9)));
                MAY(lbl_column_transmission(col, due.data(), Iin, 0.0, I, coin() ? (double*)host : nullptr, urand(1, 4)));
                MAY(lbl_column_set_layer(col, urand(0, nl - 1), c_lines.data(), c_iso.data(), &c_grid[0], c_conc.data(), 55.0, c_k[0]));
                REFUSED(lbl_column_set_layer(col, nl, c_lines.data(), c_iso.data(), &c_grid[0], c_conc.data(), 55.0, c_k[0]));
                REFUSED(lbl_column_transmission(col, nullptr, nullptr, 288.0, I, (double*)host, 0));
                MAY(lbl_column_transmission(col, nullptr, nullptr, 288.0, I, (double*)host, 3));
                MUST(lbl_download_wait(ctx));
                if (coin(0.2)) REFUSED(lbl_ctx_destroy(ctx));                   // live objects: the context stays
                MUST(lbl_column_destroy(col));
                MUST(lbl_host_free(ctx, host));
            }
            ++n_calls;
            // the same column from per-line-list cross sections (term list in device memory)
            std::vector<lbl_buffer*> c_xs;
            for (int l = 0; l < nl; ++l) for (int i = 0; i < n_lists; ++i) c_xs.push_back(xs[(size_t)i]);
            std::vector<double> c_P;
            for (int l = 0; l < nl; ++l) c_P.push_back(c.P * std::pow(0.6, l));
            MAY(lbl_column_step_dev(ctx, nl, c_niso.data(), c_xs.data(), c_isomol.data(), c_nmol.data(), c_conc.data(), c_P.data(), c_T.data(), c_depth.data(),
                                    g.range_min, g.range_max, n, 0, 0, nullptr, 288.0, c_k.data(), c_tr.data(), I));
            std::vector<lbl_buffer*> all_tr;
            for (int l = 0; l < nl; ++l) all_tr.push_back(c_tr[(size_t)l] ? c_tr[(size_t)l] : tr);
            MAY(lbl_column_sweep_dev(ctx, nl, all_tr.data(), c_T.data(), g.range_min, g.range_max, n, 0, 0, nullptr, 288.0, I));
        }
        // small entry points
        MAY(lbl_sum_dev(ctx, std::min(n_lists, 64), xs.data(), n, k));
        REFUSED(lbl_sum_dev(ctx, 65, xs.data(), n, k));
        MAY(lbl_optical_dev(ctx, tr, n, urand(0, 2), k));
        MAY(lbl_planck_dev(ctx, g.range_min, g.range_max, n, 288.0, k));
        double band = 0.0;
        MAY(lbl_band_integral(ctx, I, n, 3.14159, g.base_resolution, &band));
        MAY(lbl_line_survey_dev(ctx, L[0], &g, k));
        {
            int64_t nl0 = 0;
            MUST(lbl_lines_count(L[0], &nl0));
            std::vector<int64_t> idx((size_t)std::max<int64_t>(nl0, 1));
            std::vector<double> a((size_t)std::max<int64_t>(nl0, 1)), b(a), d(a);
            std::vector<int32_t> reg((size_t)std::max<int64_t>(nl0, 1));
            MAY(lbl_line_quantities(ctx, L[0], &iso[0], &g, idx.data(), a.data(), b.data(), d.data(), reg.data()));
        }
        if (g.shard_count == 0) {      // host arrays in and out
            std::vector<double> f((size_t)200), out((size_t)std::max<int64_t>(n, 1));
            for (size_t i = 0; i < f.size(); ++i) f[i] = g.range_min + (g.range_max - g.range_min) * (double)i / 200.0;
            int64_t rc3[3];
            MAY(lbl_xsec_accumulate(ctx, f.data(), f.data(), f.data(), f.data(), f.data(), f.data(), f.data(), 200, &iso[0], &g, out.data(), rc3));
        }
        // a step as a graph; stale after an option change
        if (coin(0.4)) {
            MAY(lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data()));
            if (lbl_capture_begin(ctx) == LBL_OK) {
                const int rc = lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data());
                lbl_graph* gr = nullptr;
                const int rc2 = lbl_capture_end(ctx, &gr);
                if (rc == LBL_OK && rc2 == LBL_OK && gr) {
                    MAY(lbl_graph_launch(gr));
                    MUST(lbl_set_option(ctx, "accuracy", 1)); MUST(lbl_set_option(ctx, "accuracy", 0));
                    MAY(lbl_graph_launch(gr));                                   // stale or not: no crash
                    MUST(lbl_graph_destroy(gr));
                } else if (gr) MUST(lbl_graph_destroy(gr));
            }
            ++n_calls;
        }
        // shard layouts back to grid order
        if (n >= 16) {
            const int world = urand(1, 8);
            const int64_t slot = (n + world - 1) / world + urand(0, 3);
            std::vector<int64_t> first((size_t)world), count((size_t)world);
            int64_t at = 0;
            for (int r = 0; r < world; ++r) { first[(size_t)r] = at; count[(size_t)r] = std::min<int64_t>(slot, n - at); if (count[(size_t)r] < 0) count[(size_t)r] = 0; at += count[(size_t)r]; }
            lbl_buffer* gathered = buffer(slot * world);
            MAY(lbl_gather_compact_dev(ctx, gathered, world, slot, first.data(), count.data(), k));
            count[0] = slot + 1;
            REFUSED(lbl_gather_compact_dev(ctx, gathered, world, slot, first.data(), count.data(), k));
        }
        // allocation failures anywhere inside a batch: a status, no leak, no crash; the context keeps working afterwards
        if (coin(0.5)) {
            mockhip::fail_malloc_at = urand(0, 6);
            MAY(lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data()));
            MAY(lbl_layer_merged_step_dev(ctx, n_lists, L.data(), iso.data(), &g, iso_mol.data(), n_mol, conc.data(), 10.0, nullptr, 288.0, k, tr, I));
            mockhip::fail_malloc_at = -1;
            default_options(ctx);
            MAY(lbl_xsec_accumulate_dev(ctx, n_lists, L.data(), iso.data(), grids.data(), xs.data()));
        }
        if (coin(0.3)) default_options(ctx);
        MUST(lbl_sync(ctx));
    }
    for (int kind = 0; kind < 6; ++kind) { int64_t la = 0; double ms = 0; MAY(lbl_profile_read(ctx, kind, &la, &ms)); }
    // teardown in a random order: views before the lists they window, everything before the context
    if (coin(0.3)) REFUSED(lbl_ctx_destroy(ctx));
    std::shuffle(bufs.begin(), bufs.end(), rng);
    std::shuffle(views.begin(), views.end(), rng);
    std::shuffle(lists.begin(), lists.end(), rng);
    const bool views_first = coin();
    if (!views_first) for (auto* b : bufs) MUST(lbl_buffer_destroy(b));
    for (auto& v : views) MUST(lbl_lines_destroy(v.h));
    for (auto& l : lists) MUST(lbl_lines_destroy(l.h));
    if (views_first) for (auto* b : bufs) MUST(lbl_buffer_destroy(b));
    MUST(lbl_ctx_destroy(ctx));
    (void)round;
}

int main(int argc, char** argv) {
    const unsigned long long seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    rng.seed(seed);
    if (lbl_abi_version() != LBL_ABI_VERSION) return 4;
    int64_t lim = 0;
    if (lbl_limit("merged_lists_per_job", &lim) != LBL_OK || lim != 64 || lbl_limit("nope", &lim) == LBL_OK) return 4;
    mockhip::device_count = 0;
    { lbl_ctx* c = nullptr; if (lbl_ctx_create(0, &c) != LBL_ERR_NO_DEVICE) return 5; }      // no device: loud, nothing half-made
    mockhip::device_count = 1;
    for (int r = 0; r < rounds; ++r) one_round(r);
    if (mockhip::live_allocs || mockhip::live_streams || mockhip::live_events || mockhip::live_graphs || mockhip::live_host) {
        fprintf(stderr, "leaked: %lld device blocks, %lld streams, %lld events, %lld graphs, %lld page-locked blocks\n", mockhip::live_allocs,
                mockhip::live_streams, mockhip::live_events, mockhip::live_graphs, mockhip::live_host);
        return 6;
    }
    if (getenv("SHIM_SITES"))
        for (int i = 0; i < 1024; ++i) if (g_site_ok[i] || g_site_no[i]) printf("line %4d: %6lld accepted %6lld refused\n", i, g_site_ok[i], g_site_no[i]);
    printf("seed %llu: %d rounds, %lld calls through the C ABI (%lld refused), no sanitizer report, nothing leaked\n", seed, rounds, n_calls, n_refused);
    return 0;
}
