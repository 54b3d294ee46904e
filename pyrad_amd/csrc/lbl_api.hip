// C-ABI of the MI355X line-by-line engine (include/pyrad_hip.h): contexts, device objects,
// argument checking and launch sequencing.  Kernels live in lbl_kernels.hip.
#include "../../include/pyrad_hip.h"
#include "lbl_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>


using namespace lbl;
namespace lbl {
void comm_quiesce(lbl_ctx* ctx);      // lbl_comm.hip: wait for unfenced collectives on buffers of ctx
void comm_forget(lbl_ctx* ctx);       // lbl_comm.hip: ctx is being destroyed
}

// ----------------------------------------------------------------------------------------
// objects
// ----------------------------------------------------------------------------------------
constexpr int kProfileKinds = 6;
enum { PROF_PREP = 0, PROF_ACCUM = 1, PROF_REGRID = 2, PROF_SWEEP = 3, PROF_COLUMN = 4, PROF_GATHER = 5 };

struct DeviceArena {      // grow-only device scratch
    void* ptr = nullptr;
    size_t cap = 0;
};

struct lbl_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;       // lbl_buffer_download_async: device-to-host copies beside later kernels
    hipEvent_t copy_ev = nullptr;
    std::string err;
    int n_cu = 0;
    // scratch
    DeviceArena recs;       // HotRec per line of the current batch
    DeviceArena cold;       // ColdRec per line
    DeviceArena cidx;       // int32 per line
    DeviceArena work;       // work grids that need a regrid
    DeviceArena jobs;       // PrepJob[] + AccumJob[] + regime counters (3 x u64 per job)
    DeviceArena counts;     // per-block regime counts of the last batch
    DeviceArena bal;        // balanced variant: span table, counts, prefix, slab
    DeviceArena red;        // band-integral partials + result
    DeviceArena zeros;      // an array of zeros: the cross section of a column layer without line lists
    size_t zeros_set = 0;
    void* host_stage = nullptr;   // pinned staging ring for job descriptors
    size_t host_stage_cap = 0;
    size_t host_stage_head = 0;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};   // one per half of the ring: "every copy out of this half has been enqueued"
    bool stage_ev_set[2] = {false, false};
    struct DescSlot { std::vector<char> bytes; void* dptr = nullptr; size_t cap = 0; };
    DescSlot desc_cache[4];
    int desc_next = 0;
    std::vector<char> desc_build;
    int last_jobs = 0;
    int last_blocks_per_job = 0;
    std::vector<int> last_list_blocks;   // blocks of 256 (lines, or merged positions of its job) each list of the last batch was counted in
    // tuning knobs (lbl_set_option)
    int accum_variant = 5;   // 0: IEEE divide + exp per pair; 1: running fraction; 2: + Gaussian recurrence
                             // (0-2 fetch records through the scalar cache); 3: 2 with wave-private LDS
                             // staging (every pair evaluated directly); 4: 3 with the balanced single-round
                             // partition of (span, line) pairs (measured: same main-kernel time as 3 plus
                             // ~20 us of helper kernels); 5 (default): 3 with the far-field series for
                             // Lorentz lines more than 4 half-spans from a span
    int accum_R = 0;         // points per lane, 0 = choose per launch
    int accum_LS = 0;        // waves sharing one span of points (line split), 0 = choose per launch
    int gauss_run = 0;       // points per lane of a Gaussian run in the far-field kernel's production shape: 0 = by the launch's wave count, 16, 32
    int bal_workers[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // resident wavefronts of the balanced kernel per R (cached)
    // schedule cache: (job, tile) lists sorted longest first, per launch group
    struct Schedule {
        std::vector<uint64_t> key; int2* d_list; int total; int32_t* d_tabs; std::vector<size_t> tab_off;
        int xcd_tol = 3;
        int sr_chunks = 1;                    // single-round launches packed per XCD: contiguous runs of tiles per XCD
        int launch_total = 0;                 // entries of d_list = workgroups of the launch (>= total: see sched_launch_items)
        int xcd_pack = 1;
        void* d_block = nullptr;              // the one allocation d_list and d_tabs live in
        // device build (launch_schedule_build): enqueued by the first batch that uses the schedule, right after its
        // line prep; until then d_list / d_tabs are uninitialised
        bool pending = false;
        int R = 0, spans_per_tile = 0, total_spans = 0, xcd_chunks = 32;
        long long far_reach = 0;
        double cost_near = 0, cost_edge = 0, cost_far = 0, cost_fixed = 0;
        std::vector<int32_t> span_first, tile_first;
        // merged layer jobs (several line lists in one record array): which line of which list belongs at every merged
        // position (the inverse of the stable merge of the lists' centre indices), built on the device by the first batch
        // that uses the schedule (launch_merge_ranks), ahead of its line prep, which runs in that order
        int32_t* d_src = nullptr;
        std::vector<size_t> src_off;         // per job of the group: first entry of its lists' positions in d_src (SIZE_MAX: one list, no merge)
        bool merge_pending = false;
    };
    std::vector<std::unique_ptr<Schedule>> schedules;
    int accuracy = 0;        // 0 exact (default): every result as close to the reference's fp64 as the arithmetic allows (1e-14);
                             // 1 budget: <= 1e-9 relative on the absorption coefficient (north_star asks 1e-6), still fp64:
                             // 18..7 far-field series terms by distance (exact: 30..12), Gaussian cut-off at 2^-34 instead of 2^-54 of the line's
                             // Lorentz term
    int sweep_ieee = 0;      // sweeps (absorption coefficient, transmittance, Planck): 0 (default) cross section x one host-computed factor
                             // conc P / 1E4 / k / T, reciprocals by rcp + Newton, exp without the library's range tests (each result
                             // within a few 1e-16 of the other form); 1 the reference's own chain of correctly rounded divisions
                             // (k bit-identical to NumPy's crossSection * concentration * P / 1E4 / k / T on the same cross section)
    int sched_build = 1;     // 1 (default): span tables and dispatch order built on the device, in stream; 0: on the host
    DeviceArena sched;       // scratch of the device build
    DeviceArena ktmp;        // lbl_layer_merged_step_dev on a work grid that needs the regrid kernel, without an abs_coef buffer: the regridded k
    DeviceArena merge_tmp;   // merged layer jobs: per-list centre indices + list descriptors while the merged positions are built
    bool sched_refused = false;   // group_schedule: a merged layer job on a launch the device build does not cover
    uint64_t lines_serial = 0;
    int lpt = 4;             // longest-first worklist: 4 (default) = 3 + XCD-partitioned when the launch has several rounds; 3 bin-packed per CU when the launch is one round; 2 snake; 1 plain; 0 positional
    int tile_order = 1;      // 1: natural order (default; measured 8 % faster on the clustered C2 grid:
                             // all CUs work through one region together); 0: each XCD gets a contiguous run
    // Graph capture (lbl_capture_begin / lbl_capture_end): while `capturing`, entry points only enqueue
    // kernels; anything that would allocate, copy or synchronise fails with LBL_ERR_STATE ("run the
    // sequence once before capturing it").  `epoch` counts every event that can invalidate a pointer a
    // captured kernel node holds (arena growth, a descriptor / argument slot being rewritten, a schedule
    // being evicted); a graph remembers the epoch it was captured at and refuses to launch after a change.
    bool capturing = false;
    uint64_t epoch = 0;
    struct ArgSlot { std::vector<char> bytes; void* dptr = nullptr; size_t cap = 0; uint64_t used = 0; };
    std::vector<ArgSlot> arg_cache;      // device copies of kernel argument blocks, found again by content
    uint64_t arg_clock = 0;
    int skew = 1;            // line lists whose window has no far line (narrower than 5 half-spans of 128 points): 1 (default) the
                             // skewed-range kernel when they fill the chip; 0 the all-direct span kernel; 2 EVERY job through the
                             // skewed-range kernel whatever its window and the grid size (parity tests)
    int xcd_tol = 3;         // ... as long as no XCD's busiest CU carries more than this many percent above the mean of the eight (-1: always)
    int xcd_pack = 1;        // single-round launches: every XCD packs its own tiles into its own CUs - 1 (default) a contiguous run of the
                             // sequence where every wave owns a span, else every 8th tile of the longest-first order; 2 / 3 always the
                             // run / always the mix; 0 one packing over all CUs by one wave (round 4)
    int xcd_chunks = 0;      // XCD-partitioned worklist: contiguous chunks of the tile sequence per XCD (0: by the workgroup count, 10..32)
    int skew_LS = 0;         // waves sharing a span in the skewed-range kernel: 0 (auto: by the lines per point), 1, 2, 4
    int skew_R = 8;          // points per lane of the skewed-range kernel (8: 118 VGPRs, 4 waves per SIMD; measured 7 % faster than 4 on the column)
    int far_min_H = 0;       // windows below this many points go to the skewed-range kernel even if they have far lines (0: the far-field kernel's own limit, 640)
    int ablate = 0;          // LBL_DIAG builds: AccumJob.ablate (always 0 in the production library)
    const PrepJob* last_prep_desc = nullptr;   // device copy of the last batch's PrepJob[] (lbl_line_quantities reads job 0)
    lbl_ctx* chain_pred = nullptr;   // lbl_ctx_chain_accumulate: accumulate kernels wait for this context's
    hipEvent_t accum_done = nullptr; // recorded after this context's accumulate launches
    bool no_fuse = false;    // lbl_layer_step_dev as accumulate + separate sweep launch (A/B, parity tests)
    int live_objects = 0;
    // event timing (lbl_profile_*)
    unsigned profiling = 0;        // bit k set: time kernel class k
    std::vector<hipEvent_t> ev_pool;                       // idle events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_rec[kProfileKinds];
};

struct lbl_buffer {
    lbl_ctx* ctx;
    double* d;
    int64_t n;
};

struct lbl_lines {
    lbl_ctx* ctx;
    double* d;        // 7 arrays of `stride` doubles: nu, sw, elower, gamma_air, gamma_self, n_air, delta_air (a view: offset into its root's)
    int64_t n;
    std::vector<double> host_nu_own;   // for scheduling only: longest-first tile order (never used for results)
    const double* host_nu = nullptr;   // ... the owning list's copy; a view points into its root's (which outlives it: `views`)
    uint64_t serial;               // identity for the schedule cache
    int64_t stride = 0;            // distance between the field arrays: the OWNING list's line count
    lbl_lines* root = nullptr;     // a view (lbl_lines_view): the list that owns the device arrays
    int views = 0;                 // live views of this (owning) list
    const double* field(int k) const { return d + (size_t)k * (size_t)stride; }
};

static thread_local std::string g_err;
// every live context of the process: lbl_ctx_destroy unlinks the dying one from contexts chained to it
static std::mutex g_ctx_mutex;
static std::vector<lbl_ctx*> g_contexts;

// LBL_TRACE=1: host-side phase timings on stderr (setup paths only: allocation, upload, schedule build)
static const bool g_trace = getenv("LBL_TRACE") != nullptr;
struct TraceScope {
    const char* what; long long n; std::chrono::steady_clock::time_point t0;
    TraceScope(const char* w, long long n_ = 0) : what(w), n(n_) { if (g_trace) t0 = std::chrono::steady_clock::now(); }
    ~TraceScope() {
        if (g_trace)
            fprintf(stderr, "[lbl trace] %-28s %10.1f us  (%lld)\n", what,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), n);
    }
};

static int fail(lbl_ctx* ctx, int code, const char* fmt, ...) noexcept {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try {
        if (ctx) ctx->err = buf; else g_err = buf;
    } catch (...) {
        // the message is lost, the status code is not
    }
    return code;
}

// No C++ exception crosses the C boundary (include/pyrad_hip.h): every extern "C" entry point is a
// function-try-block closed by this handler list.  Host containers sized by caller input
// (std::vector in enqueue_accumulate / group_schedule / lbl_last_regime_counts ...) throw
// bad_alloc or length_error on absurd sizes: both are reported as LBL_ERR_OOM.
#define LBL_GUARD_END(ctx_expr)                                                                          \
    catch (const std::bad_alloc&) { return fail((ctx_expr), LBL_ERR_OOM, "%s: host allocation failed", __func__); }      \
    catch (const std::length_error&) { return fail((ctx_expr), LBL_ERR_OOM, "%s: host allocation too large", __func__); } \
    catch (const std::exception& e) { return fail((ctx_expr), LBL_ERR_STATE, "%s: %s", __func__, e.what()); }             \
    catch (...) { return fail((ctx_expr), LBL_ERR_STATE, "%s: unknown C++ exception", __func__); }

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, e_ == hipErrorOutOfMemory ? LBL_ERR_OOM : LBL_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

static int capture_refuses(lbl_ctx* ctx, const char* what) {
    return fail(ctx, LBL_ERR_STATE, "graph capture: %s is not possible while capturing - run the same sequence once "
                                    "before lbl_capture_begin so that buffers, schedules and descriptors exist", what);
}

// (the sanitizer harness of this file, tests/host_shim, builds with -DLBL_SANITIZER_BUILD: scratch blocks then carry no slack, so
// that AddressSanitizer sees a block whose need was computed too small instead of a write into the slack)
#ifdef LBL_SANITIZER_BUILD
static size_t block_slack(size_t, size_t) { return 0; }
#else
static size_t block_slack(size_t bytes, size_t fixed) { return bytes / 4 + fixed; }
#endif

static int arena_reserve(lbl_ctx* ctx, DeviceArena& a, size_t bytes) {
    if (bytes <= a.cap) return LBL_OK;
    if (ctx->capturing) return capture_refuses(ctx, "growing a scratch buffer");
    TraceScope tr("arena grow", (long long)bytes);
    ctx->epoch++;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (a.ptr) HIP_TRY(ctx, hipFree(a.ptr));
    a.ptr = nullptr; a.cap = 0;
    size_t want = bytes + block_slack(bytes, 4096);
    HIP_TRY(ctx, hipMalloc(&a.ptr, want));
    a.cap = want;
    return LBL_OK;
}

// RAII-free scoped timer: prof_begin records the start event, prof_end the stop event.
static hipEvent_t prof_event(lbl_ctx* ctx) {
    hipEvent_t e = nullptr;
    if (!ctx->ev_pool.empty()) { e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
static hipEvent_t prof_begin(lbl_ctx* ctx, int kind) {
    if (ctx->capturing || !((ctx->profiling >> kind) & 1u)) return nullptr;
    hipEvent_t e = prof_event(ctx);
    if (e) (void)hipEventRecord(e, ctx->stream);
    return e;
}
static void prof_end(lbl_ctx* ctx, int kind, hipEvent_t start) {
    if (!start) return;
    hipEvent_t e = prof_event(ctx);
    if (!e) { ctx->ev_pool.push_back(start); return; }
    (void)hipEventRecord(e, ctx->stream);
    ctx->ev_rec[kind].emplace_back(start, e);
}

// Pinned staging for descriptors that a later hipMemcpyAsync reads: bump-allocated from a ring of two halves.  A half is
// reused only after the copies that read it have run: leaving a half records an event on the stream (every copy out of that
// half was enqueued before it), entering a half waits for ITS event - recorded half a ring of descriptors ago, so normally long
// complete.  (Until round 6 a wrap drained the whole stream: a step with fresh descriptors - a re-windowed column, whose
// blocks are ~150 KB per call - then stalled for everything it had just enqueued, 7 ms of a 14 ms call every sixth call or so.)
static int stage_alloc(lbl_ctx* ctx, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (ctx->capturing) return capture_refuses(ctx, "staging a host-to-device copy");
    if (bytes * 8 > ctx->host_stage_cap) {                 // (an allocation never exceeds an eighth of the ring: a quarter of a half)
        TraceScope tr("staging ring grow", (long long)bytes);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->host_stage) HIP_TRY(ctx, hipHostFree(ctx->host_stage));
        ctx->host_stage = nullptr; ctx->host_stage_cap = 0; ctx->host_stage_head = 0;
        ctx->stage_ev_set[0] = ctx->stage_ev_set[1] = false;
#ifdef LBL_SANITIZER_BUILD
        size_t want = bytes * 8;                               // (the harness wants the ring to wrap often)
#else
        size_t want = std::max<size_t>(bytes * 8, (size_t)1 << 20);
#endif
        HIP_TRY(ctx, hipHostMalloc(&ctx->host_stage, want, hipHostMallocDefault));
        ctx->host_stage_cap = want;
    }
    const size_t half = ctx->host_stage_cap / 2;
    const int from = ctx->host_stage_head < half ? 0 : 1;
    size_t head = ctx->host_stage_head;
    if (head < half && head + bytes > half) head = half;                    // no allocation straddles the two halves
    if (head + bytes > ctx->host_stage_cap) head = 0;
    const int to = head < half ? 0 : 1;
    if (to != from) {
        for (int h = 0; h < 2; ++h)
            if (!ctx->stage_ev[h]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->stage_ev[h], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->stage_ev[from], ctx->stream));
        ctx->stage_ev_set[from] = true;
        if (ctx->stage_ev_set[to]) {
            TraceScope tr("staging ring: wait for a half", (long long)to);
            HIP_TRY(ctx, hipEventSynchronize(ctx->stage_ev[to]));
        }
    }
    *out = (char*)ctx->host_stage + head;
    ctx->host_stage_head = head + bytes;
    return LBL_OK;
}

// Device copy of a kernel argument block (column sweeps): found again by content, so a step that
// repeats uploads nothing - which is also what lets it be captured into a graph.
static int device_args(lbl_ctx* ctx, const void* host, size_t bytes, void** dptr) {
    const char* hb = (const char*)host;
    for (auto& e : ctx->arg_cache)
        if (e.dptr && e.bytes.size() == bytes && !memcmp(e.bytes.data(), hb, bytes)) {
            e.used = ++ctx->arg_clock;
            *dptr = e.dptr;
            return LBL_OK;
        }
    if (ctx->capturing) return capture_refuses(ctx, "uploading kernel arguments");
    lbl_ctx::ArgSlot* slot = nullptr;
    if (ctx->arg_cache.size() < 8) {
        ctx->arg_cache.emplace_back();
        slot = &ctx->arg_cache.back();
    } else {
        slot = &ctx->arg_cache[0];
        for (auto& e : ctx->arg_cache) if (e.used < slot->used) slot = &e;     // least recently used
        ctx->epoch++;                                                           // a captured graph may point at it
    }
    if (slot->cap < bytes) {
        TraceScope tr("argument slot alloc", (long long)bytes);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (slot->dptr) HIP_TRY(ctx, hipFree(slot->dptr));
        slot->dptr = nullptr; slot->cap = 0; slot->bytes.clear();
        const size_t want = bytes + (block_slack(0, 256));
        HIP_TRY(ctx, hipMalloc(&slot->dptr, want));
        slot->cap = want;
    }
    void* pinned = nullptr;
    int rc = stage_alloc(ctx, bytes, &pinned);
    if (rc) return rc;
    memcpy(pinned, host, bytes);
    HIP_TRY(ctx, hipMemcpyAsync(slot->dptr, pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    slot->bytes.assign(hb, hb + bytes);
    slot->used = ++ctx->arg_clock;
    *dptr = slot->dptr;
    return LBL_OK;
}

static int check_grid(lbl_ctx* ctx, const lbl_grid* g) {
    if (!g) return fail(ctx, LBL_ERR_BAD_ARG, "grid is NULL");
    if (!(g->resolution > 0) || !(g->base_resolution > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "resolution must be > 0");
    if (g->n_work < 0 || g->n_base < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative grid length");
    if (g->window < 1) return fail(ctx, LBL_ERR_BAD_ARG, "window (len(arange(0,dfc,res))) must be >= 1: the reference raises IndexError at rightCurve[0]");
    if (g->n_work > 1900000000LL || g->window > 100000000LL) return fail(ctx, LBL_ERR_BAD_ARG, "grid too large for int32 indexing");
    if (!(g->range_max >= g->range_min)) return fail(ctx, LBL_ERR_BAD_ARG, "range_max < range_min");
    if (g->shard_first < 0 || g->shard_count < 0 || g->shard_first + g->shard_count > g->n_work)
        return fail(ctx, LBL_ERR_BAD_ARG, "shard outside the work grid");
    if (g->shard_count > 0 && !(g->resolution == g->base_resolution && g->n_work == g->n_base))
        return fail(ctx, LBL_ERR_BAD_ARG, "a sharded accumulate needs resolution == base_resolution (regrid after the gather)");
    return LBL_OK;
}

static void shard_range(const lbl_grid& g, long long* first, long long* count) {
    if (g.shard_count > 0) { *first = g.shard_first; *count = g.shard_count; }
    else { *first = 0; *count = g.n_work; }
}

static bool needs_regrid(const lbl_grid& g) {
    return !(g.resolution == g.base_resolution && g.n_work == g.n_base);
}

// ----------------------------------------------------------------------------------------
// library / context
// ----------------------------------------------------------------------------------------
extern "C" int lbl_abi_version(void) { return LBL_ABI_VERSION; }

// The library's fixed sizes, for hosts that pick a route by them (pyrad_amd.model: merged layer step or per-line-list step)
extern "C" int lbl_limit(const char* name, int64_t* value) {
    if (!name || !value) return LBL_ERR_BAD_ARG;
    if (!strcmp(name, "merged_lists_per_job")) *value = kMaxIso;               // line lists of one merged layer job
    else if (!strcmp(name, "arrays_per_layer")) *value = kMaxColumnIso - 1;    // cross-section arrays lbl_layer_sweep_dev / lbl_layer_step_dev take
    else if (!strcmp(name, "arrays_per_sum")) *value = kMaxIso;                // inputs of lbl_sum_dev
    else if (!strcmp(name, "arrays_per_column")) *value = kMaxColumnIso - 1;   // terms of lbl_column_step_dev
    else if (!strcmp(name, "layers_per_column")) *value = kMaxLayers;
    else if (!strcmp(name, "jobs_per_batch")) *value = LBL_MAX_JOBS;
    else return LBL_ERR_BAD_ARG;
    return LBL_OK;
}

extern "C" int lbl_device_count(int* count) try {
    if (!count) return fail(nullptr, LBL_ERR_BAD_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, LBL_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return LBL_OK;
} LBL_GUARD_END(nullptr)

extern "C" int lbl_ctx_create(int device, lbl_ctx** out) try {
    if (!out) return fail(nullptr, LBL_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, LBL_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, LBL_ERR_NO_DEVICE, "device %d out of range [0,%d)", device, n);
    lbl_ctx* ctx = new (std::nothrow) lbl_ctx();
    if (!ctx) return fail(nullptr, LBL_ERR_OOM, "host allocation failed");
    ctx->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { int rc = fail(nullptr, LBL_ERR_HIP, "context setup: %s", hipGetErrorString(e)); delete ctx; return rc; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
    { std::lock_guard<std::mutex> lock(g_ctx_mutex); g_contexts.push_back(ctx); }
    *out = ctx;
    return LBL_OK;
} LBL_GUARD_END(nullptr)

extern "C" int lbl_ctx_destroy(lbl_ctx* ctx) try {
    if (!ctx) return LBL_OK;
    if (ctx->live_objects != 0) return fail(ctx, LBL_ERR_STATE, "%d device objects still alive", ctx->live_objects);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    {   // a successor chained to this context (lbl_ctx_chain_accumulate) must not wait on its event any more
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), ctx), g_contexts.end());
        for (lbl_ctx* c : g_contexts) if (c->chain_pred == ctx) c->chain_pred = nullptr;
    }
    lbl::comm_forget(ctx);
    if (ctx->accum_done) (void)hipEventDestroy(ctx->accum_done);
    for (auto& v : ctx->ev_rec) for (auto& p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto& sc : ctx->schedules) if (sc->d_block) (void)hipFree(sc->d_block);
    for (auto& e : ctx->desc_cache) if (e.dptr) (void)hipFree(e.dptr);
    for (auto& e : ctx->arg_cache) if (e.dptr) (void)hipFree(e.dptr);
    DeviceArena* arenas[] = {&ctx->recs, &ctx->cold, &ctx->cidx, &ctx->work, &ctx->jobs, &ctx->counts, &ctx->bal, &ctx->red, &ctx->zeros, &ctx->sched, &ctx->merge_tmp, &ctx->ktmp};
    for (DeviceArena* a : arenas) if (a->ptr) (void)hipFree(a->ptr);
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    for (int h = 0; h < 2; ++h) if (ctx->stage_ev[h]) (void)hipEventDestroy(ctx->stage_ev[h]);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" const char* lbl_last_error(const lbl_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

extern "C" int lbl_sync(lbl_ctx* ctx) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_ctx_stream(lbl_ctx* ctx, void** stream) try {
    if (!ctx || !stream) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *stream = (void*)ctx->stream;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_ctx_chain_accumulate(lbl_ctx* ctx, lbl_ctx* predecessor) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (predecessor && predecessor->device != ctx->device) return fail(ctx, LBL_ERR_BAD_ARG, "contexts live on different devices");
    if (predecessor == ctx) return fail(ctx, LBL_ERR_BAD_ARG, "a context cannot follow itself");
    if (ctx->capturing || (predecessor && predecessor->capturing)) return fail(ctx, LBL_ERR_STATE, "cannot chain a context while it is capturing a graph");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (lbl_ctx* c : {ctx, predecessor})
        if (c && !c->accum_done) HIP_TRY(ctx, hipEventCreateWithFlags(&c->accum_done, hipEventDisableTiming));
    ctx->chain_pred = predecessor;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_device_info(lbl_ctx* ctx, char* name, int name_len, int* n_cu, int64_t* hbm_bytes) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s %s", prop.name, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_profile_enable(lbl_ctx* ctx, int on) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    ctx->profiling = on > 0 ? (unsigned)on : 0u;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_profile_reserve(lbl_ctx* ctx, int n_events) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->ev_pool.size() < n_events) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_profile_reset(lbl_ctx* ctx) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& v : ctx->ev_rec) {
        for (auto& p : v) { ctx->ev_pool.push_back(p.first); ctx->ev_pool.push_back(p.second); }
        v.clear();
    }
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_profile_read(lbl_ctx* ctx, int kind, int64_t* launches, double* total_ms) try {
    if (!ctx || !launches || !total_ms) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (kind < 0 || kind >= kProfileKinds) return fail(ctx, LBL_ERR_BAD_ARG, "kind out of range");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (auto& p : ctx->ev_rec[kind]) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, p.first, p.second));
        tot += ms;
    }
    *launches = (int64_t)ctx->ev_rec[kind].size();
    *total_ms = tot;
    return LBL_OK;
} LBL_GUARD_END(ctx)

// hook for lbl_comm.hip: time the all-gather like any other kernel class
namespace lbl {
void* comm_prof_begin(lbl_ctx* ctx) { return (void*)prof_begin(ctx, PROF_GATHER); }
void comm_prof_end(lbl_ctx* ctx, void* start) { prof_end(ctx, PROF_GATHER, (hipEvent_t)start); }
}  // namespace lbl

// Tuning knobs for benchmarking and A/B parity runs (not part of the reference surface):
//   "accum_variant" 0 | 1 | 2,  "accum_points_per_lane" 0 (auto) | 1 | 2 | 4 | 8
// Every option decides which kernels, launch shapes or arithmetic the batches enqueued afterwards use; a graph captured
// before (lbl_capture_end) would go on replaying the old ones.  A change of any option therefore bumps the context's epoch:
// lbl_graph_launch reports the graph stale (LBL_ERR_STATE) and the caller captures again (engine.StepGraph does by itself).
static uint64_t option_state(const lbl_ctx* c) {
    const long long v[] = {c->accum_variant, c->accum_R, c->accum_LS, c->gauss_run, c->lpt, c->tile_order, c->skew, c->skew_R, c->skew_LS, c->xcd_chunks, c->xcd_pack, c->xcd_tol, c->far_min_H,
                           c->ablate, c->accuracy, c->sweep_ieee, c->sched_build, c->no_fuse ? 1 : 0,
                           c->bal_workers[1], c->bal_workers[2], c->bal_workers[4], c->bal_workers[8]};
    uint64_t h = 1469598103934665603ull;
    for (long long x : v) { h ^= (uint64_t)x; h *= 1099511628211ull; }
    return h;
}
static int set_option_value(lbl_ctx* ctx, const char* key, int value);

extern "C" int lbl_set_option(lbl_ctx* ctx, const char* key, int value) try {
    if (!ctx || !key) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    const uint64_t before = option_state(ctx);
    const int rc = set_option_value(ctx, key, value);
    if (option_state(ctx) != before) ctx->epoch++;
    return rc;
} LBL_GUARD_END(ctx)

static int set_option_value(lbl_ctx* ctx, const char* key, int value) {
    if (!strcmp(key, "accum_variant")) {
#ifdef LBL_DIAG
        if (value < 0 || value > 5) return fail(ctx, LBL_ERR_BAD_ARG, "accum_variant must be 0..5");
#else
        if (!(value == 0 || value == 3 || value == 5))
            return fail(ctx, LBL_ERR_BAD_ARG, "accum_variant must be 0, 3 or 5 (1, 2 and 4 are superseded comparison kernels: diagnostic builds only)");
#endif
        ctx->accum_variant = value;
    } else if (!strcmp(key, "accum_points_per_lane")) {
        if (!(value == 0 || value == 1 || value == 2 || value == 4 || value == 8))
            return fail(ctx, LBL_ERR_BAD_ARG, "accum_points_per_lane must be 0, 1, 2, 4 or 8");
        ctx->accum_R = value;
    } else if (!strcmp(key, "accum_longest_first")) {
        if (value < 0 || value > 4) return fail(ctx, LBL_ERR_BAD_ARG, "accum_longest_first must be 0..4");
        ctx->lpt = value;
    } else if (!strcmp(key, "accum_blocks_per_cu")) {
        if (value < 0 || value > 8) return fail(ctx, LBL_ERR_BAD_ARG, "accum_blocks_per_cu must be 0 (auto) .. 8");
        for (int r = 0; r < 9; ++r) ctx->bal_workers[r] = value ? (ctx->n_cu > 0 ? ctx->n_cu : 256) * value * 4 : 0;
    } else if (!strcmp(key, "accum_tile_order")) {
        if (value < 0 || value > 2) return fail(ctx, LBL_ERR_BAD_ARG, "accum_tile_order must be 0, 1 or 2");
        ctx->tile_order = value;
    } else if (!strcmp(key, "accum_line_split")) {
        if (!(value == 0 || value == 1 || value == 2 || value == 4 || value == 8))
            return fail(ctx, LBL_ERR_BAD_ARG, "accum_line_split must be 0, 1, 2, 4 or 8");
        ctx->accum_LS = value;
    } else if (!strcmp(key, "accum_skew")) {
        if (value < 0 || value > 2) return fail(ctx, LBL_ERR_BAD_ARG, "accum_skew must be 0, 1 or 2");
        ctx->skew = value;
    } else if (!strcmp(key, "accum_skew_points_per_lane")) {
        if (!(value == 1 || value == 2 || value == 4 || value == 8))
            return fail(ctx, LBL_ERR_BAD_ARG, "accum_skew_points_per_lane must be 1, 2, 4 or 8");
        ctx->skew_R = value;
    } else if (!strcmp(key, "accum_xcd_chunks")) {
        if (value < 0 || value > 64) return fail(ctx, LBL_ERR_BAD_ARG, "accum_xcd_chunks must be 0 (auto: 10..32 by the workgroup count) .. 64");
        ctx->xcd_chunks = value;
    } else if (!strcmp(key, "accum_xcd_tolerance")) {
        if (value < -1 || value > 15) return fail(ctx, LBL_ERR_BAD_ARG, "accum_xcd_tolerance must be -1 (off) .. 15 percent");
        ctx->xcd_tol = value;
    } else if (!strcmp(key, "accum_xcd_pack")) {
        if (value < 0 || value > 3) return fail(ctx, LBL_ERR_BAD_ARG, "accum_xcd_pack must be 0..3");
        ctx->xcd_pack = value;
    } else if (!strcmp(key, "accum_skew_line_split")) {
        if (!(value == 0 || value == 1 || value == 2 || value == 4))
            return fail(ctx, LBL_ERR_BAD_ARG, "accum_skew_line_split must be 0 (auto), 1, 2 or 4");
        ctx->skew_LS = value;
    } else if (!strcmp(key, "accum_gauss_run")) {
        if (value != 0 && value != 16 && value != 32) return fail(ctx, LBL_ERR_BAD_ARG, "accum_gauss_run must be 0 (auto), 16 or 32");
        ctx->gauss_run = value;
    } else if (!strcmp(key, "accum_far_min_window")) {
        if (value < 0) return fail(ctx, LBL_ERR_BAD_ARG, "accum_far_min_window must be >= 0");
        ctx->far_min_H = value;
#ifdef LBL_DIAG
    } else if (!strcmp(key, "debug_ablate")) {
        ctx->ablate = value;        // diagnostic builds only, timing experiments: results are wrong when non-zero
#endif
    } else if (!strcmp(key, "accuracy")) {
        if (value < 0 || value > 1) return fail(ctx, LBL_ERR_BAD_ARG, "accuracy must be 0 (exact) or 1 (budget: 1e-9)");
        ctx->accuracy = value;
    } else if (!strcmp(key, "sweep_ieee_divisions")) {
        if (value < 0 || value > 1) return fail(ctx, LBL_ERR_BAD_ARG, "sweep_ieee_divisions must be 0 or 1");
        ctx->sweep_ieee = value;
    } else if (!strcmp(key, "schedule_build")) {
        if (value < 0 || value > 1) return fail(ctx, LBL_ERR_BAD_ARG, "schedule_build must be 0 (host) or 1 (device)");
        ctx->sched_build = value;
    } else if (!strcmp(key, "layer_step_fused")) {
        if (value < 0 || value > 1) return fail(ctx, LBL_ERR_BAD_ARG, "layer_step_fused must be 0 or 1");
        ctx->no_fuse = value == 0;
    } else if (!strcmp(key, "debug_throw")) {
        // test hook: raise inside the library to prove that the boundary holds (tests/test_gpu_abi.py)
        if (value == 1) throw std::bad_alloc();
        if (value == 2) throw std::runtime_error("debug_throw");
        if (value == 3) { std::vector<double> v; v.reserve(v.max_size() + 1); }     // length_error
    } else {
        return fail(ctx, LBL_ERR_BAD_ARG, "unknown option '%s'", key);
    }
    return LBL_OK;
}

// ----------------------------------------------------------------------------------------
// buffers
// ----------------------------------------------------------------------------------------
extern "C" int lbl_buffer_create(lbl_ctx* ctx, int64_t n, lbl_buffer** out) try {
    if (!ctx || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative length");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TraceScope tr("buffer_create", (long long)n);
    double* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, (size_t)std::max<int64_t>(n, 1) * sizeof(double)));
    lbl_buffer* b = new (std::nothrow) lbl_buffer{ctx, d, n};
    if (!b) { (void)hipFree(d); return fail(ctx, LBL_ERR_OOM, "host allocation failed"); }
    ctx->live_objects++;
    *out = b;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_buffer_destroy(lbl_buffer* buf) try {
    if (!buf) return LBL_OK;
    lbl_ctx* ctx = buf->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    lbl::comm_quiesce(ctx);           // an overlapped all-gather may still read or write it on the communicator's stream
    HIP_TRY(ctx, hipFree(buf->d));
    ctx->epoch++;                     // a captured graph may hold its address (lbl_graph_launch then refuses)
    ctx->live_objects--;
    delete buf;
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

extern "C" int lbl_buffer_size(const lbl_buffer* buf, int64_t* n) try {
    if (!buf || !n) return fail(nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    *n = buf->n;
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

extern "C" int lbl_buffer_upload(lbl_buffer* buf, const double* host, int64_t n, int64_t dst_offset) try {
    if (!buf) return fail(nullptr, LBL_ERR_BAD_ARG, "buf is NULL");
    lbl_ctx* ctx = buf->ctx;
    if (n < 0 || dst_offset < 0 || dst_offset + n > buf->n) return fail(ctx, LBL_ERR_BAD_ARG, "upload range out of bounds");
    if (n == 0) return LBL_OK;
    if (!host) return fail(ctx, LBL_ERR_BAD_ARG, "host is NULL");
    HIP_TRY(ctx, hipMemcpyAsync(buf->d + dst_offset, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // the host pointer is not kept past return
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

extern "C" int lbl_buffer_download(lbl_buffer* buf, double* host, int64_t n, int64_t src_offset) try {
    if (!buf) return fail(nullptr, LBL_ERR_BAD_ARG, "buf is NULL");
    lbl_ctx* ctx = buf->ctx;
    if (n < 0 || src_offset < 0 || src_offset + n > buf->n) return fail(ctx, LBL_ERR_BAD_ARG, "download range out of bounds");
    if (n == 0) return LBL_OK;
    if (!host) return fail(ctx, LBL_ERR_BAD_ARG, "host is NULL");
    HIP_TRY(ctx, hipMemcpyAsync(host, buf->d + src_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

// Download that does not wait: the copy is ordered behind everything enqueued on the context stream so far and runs on
// the context's copy stream, beside the kernels enqueued after it (a column's outgoing spectrum leaves in pieces while the
// fold of the next piece runs).  `host` should be page-locked (lbl_host_alloc) - pageable memory makes the runtime stage
// the copy and the call blocks.  The range must not be rewritten, nor `host` read, before lbl_download_wait.
extern "C" int lbl_buffer_download_async(lbl_buffer* buf, double* host, int64_t n, int64_t src_offset) try {
    if (!buf) return fail(nullptr, LBL_ERR_BAD_ARG, "buf is NULL");
    lbl_ctx* ctx = buf->ctx;
    if (n < 0 || src_offset < 0 || n > buf->n - src_offset) return fail(ctx, LBL_ERR_BAD_ARG, "download range out of bounds");
    if (n == 0) return LBL_OK;
    if (!host) return fail(ctx, LBL_ERR_BAD_ARG, "host is NULL");
    if (ctx->capturing) return capture_refuses(ctx, "an asynchronous download");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (!ctx->copy_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->copy_ev, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->copy_ev, 0));
    HIP_TRY(ctx, hipMemcpyAsync(host, buf->d + src_offset, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->copy_stream));
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

extern "C" int lbl_download_wait(lbl_ctx* ctx) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (ctx->copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_buffer_fill(lbl_buffer* buf, double value) try {
    if (!buf) return fail(nullptr, LBL_ERR_BAD_ARG, "buf is NULL");
    lbl_ctx* ctx = buf->ctx;
    if (buf->n == 0) return LBL_OK;
    if (value == 0.0) {
        HIP_TRY(ctx, hipMemsetAsync(buf->d, 0, (size_t)buf->n * sizeof(double), ctx->stream));
    } else {
        uint64_t bits;
        memcpy(&bits, &value, 8);
        // hipMemsetD32 cannot express a 64-bit pattern with different halves; stage through the host
        std::vector<double> tmp((size_t)buf->n, value);
        HIP_TRY(ctx, hipMemcpyAsync(buf->d, tmp.data(), (size_t)buf->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

extern "C" int lbl_host_alloc(lbl_ctx* ctx, int64_t bytes, void** out) try {
    if (!ctx || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (bytes < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocDefault));
    return LBL_OK;
} LBL_GUARD_END(ctx)

// ctx may be NULL: the block outlived its context (a caller still held an array in it)
extern "C" int lbl_host_free(lbl_ctx* ctx, void* ptr) try {
    if (!ptr) return LBL_OK;
    if (ctx) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostFree(ptr));
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_buffer_devptr(lbl_buffer* buf, void** devptr) try {
    if (!buf || !devptr) return fail(nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    *devptr = (void*)buf->d;
    return LBL_OK;
} LBL_GUARD_END(buf ? buf->ctx : nullptr)

// ----------------------------------------------------------------------------------------
// line lists
// ----------------------------------------------------------------------------------------
extern "C" int lbl_lines_create(lbl_ctx* ctx, const double* nu, const double* sw, const double* elower,
                                const double* gamma_air, const double* gamma_self, const double* n_air,
                                const double* delta_air, int64_t n_lines, lbl_lines** out) try {
    if (!ctx || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (n_lines < 0 || n_lines > 2000000000LL) return fail(ctx, LBL_ERR_BAD_ARG, "bad line count");
    const double* src[7] = {nu, sw, elower, gamma_air, gamma_self, n_air, delta_air};
    if (n_lines > 0)
        for (const double* p : src) if (!p) return fail(ctx, LBL_ERR_BAD_ARG, "NULL line field");
    for (int64_t i = 1; i < n_lines; ++i)
        if (!(nu[i] >= nu[i - 1])) return fail(ctx, LBL_ERR_BAD_ARG, "nu must be non-decreasing (line %lld)", (long long)i);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TraceScope tr("lines_create", (long long)n_lines);
    // host-side allocations first: if they throw, nothing on the device has to be released
    std::vector<double> host_nu(nu, nu + n_lines);
    lbl_lines* L = new (std::nothrow) lbl_lines{ctx, nullptr, n_lines, {}, nullptr, 0};
    if (!L) return fail(ctx, LBL_ERR_OOM, "host allocation failed");
    L->host_nu_own.swap(host_nu);
    L->host_nu = L->host_nu_own.data();
    L->stride = n_lines;
    double* d = nullptr;
    {
        hipError_t em = hipMalloc((void**)&d, (size_t)std::max<int64_t>(n_lines, 1) * 7 * sizeof(double));
        if (em != hipSuccess) { delete L; return fail(ctx, em == hipErrorOutOfMemory ? LBL_ERR_OOM : LBL_ERR_HIP, "line list allocation: %s", hipGetErrorString(em)); }
    }
    L->d = d;
    for (int k = 0; k < 7 && n_lines > 0; ++k) {
        hipError_t e = hipMemcpyAsync(d + (size_t)k * n_lines, src[k], (size_t)n_lines * sizeof(double),
                                      hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { (void)hipFree(d); delete L; return fail(ctx, LBL_ERR_HIP, "line upload: %s", hipGetErrorString(e)); }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d); delete L; return fail(ctx, LBL_ERR_HIP, "line upload: %s", hipGetErrorString(e)); }
    L->serial = ++ctx->lines_serial;
    ctx->live_objects++;
    *out = L;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_lines_view(lbl_lines* parent, int64_t first, int64_t count, lbl_lines** out) try {
    if (!parent || !out) return fail(parent ? parent->ctx : nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    lbl_ctx* ctx = parent->ctx;
    *out = nullptr;
    if (first < 0 || count < 0 || first > parent->n || count > parent->n - first) return fail(ctx, LBL_ERR_BAD_ARG, "view outside the line list");
    lbl_lines* root = parent->root ? parent->root : parent;
    lbl_lines* V = new (std::nothrow) lbl_lines{ctx, parent->d + first, count, {}, nullptr, 0};
    if (!V) return fail(ctx, LBL_ERR_OOM, "host allocation failed");
    V->host_nu = parent->host_nu ? parent->host_nu + first : nullptr;        // (no copy: 1 MB and 0.1 ms per view of 131,072 lines, 90 views per re-windowed column)
    V->stride = root->stride;
    V->root = root;
    V->serial = ++ctx->lines_serial;
    root->views++;
    ctx->live_objects++;
    *out = V;
    return LBL_OK;
} LBL_GUARD_END(parent ? parent->ctx : nullptr)

extern "C" int lbl_lines_destroy(lbl_lines* lines) try {
    if (!lines) return LBL_OK;
    lbl_ctx* ctx = lines->ctx;
    if (lines->views > 0) return fail(ctx, LBL_ERR_STATE, "%d views of this line list are still alive", lines->views);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (lines->root) {                 // a view: the arrays belong to its root
        lines->root->views--;
        ctx->epoch++;
        ctx->live_objects--;
        delete lines;
        return LBL_OK;
    }
    HIP_TRY(ctx, hipFree(lines->d));
    ctx->epoch++;                     // a captured graph may hold its address (lbl_graph_launch then refuses)
    ctx->live_objects--;
    delete lines;
    return LBL_OK;
} LBL_GUARD_END(lines ? lines->ctx : nullptr)

extern "C" int lbl_lines_count(const lbl_lines* lines, int64_t* n) try {
    if (!lines || !n) return fail(nullptr, LBL_ERR_BAD_ARG, "NULL argument");
    *n = lines->n;
    return LBL_OK;
} LBL_GUARD_END(lines ? lines->ctx : nullptr)

// ----------------------------------------------------------------------------------------
// the hot path
// ----------------------------------------------------------------------------------------
// Launch shape of the LDS variant, from measurements on MI355X (scripts/sweep*.sh):
//   R  points per lane: 4 is the sweet spot (R = 8 needs 136 VGPRs: spills or a slower loop);
//      never more points per wave than a line's support is wide.
//   LS waves sharing one span and splitting its lines: 4 when a span sees >= 1024 lines,
//      2 from 256 (the four waves of a workgroup then finish together and workgroups are
//      short, which balances the clustered line density), else 1.
// The scalar-cache variants (0-2) keep their original rule: about 3 waves per SIMD.
static void choose_shape(const lbl_ctx* ctx, int variant, long long total_points, long long total_lines, long long min_H,
                         int* R_out, int* LS_out) {
    const long long cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
    const bool lds = variant >= 3;
    int R = ctx->accum_R, LS = lds ? ctx->accum_LS : 1;
    auto lines_per_span = [&](int r) {
        return total_points > 0 ? (double)total_lines / (double)total_points * (double)(2 * min_H + 64LL * r) : 0.0;
    };
    if (ctx->accum_variant == 5 && !R && !LS) {   // (also for its groups without far lines, which run variant 3's kernel)
        // far-field kernel: R = 4 and as little line split as the line count asks for, unless the grid
        // is too small to give every SIMD two wavefronts: then split more, then shrink the spans
        // (C1, 10^4 points: R = 1 with 8 waves per span is 4x faster than R = 4 unsplit)
        // Round 5 (merged layer jobs carry three times the lines per span): the split is the one a small model of the kernel
        // prefers - a span costs W = 600 + lps (1.5 + 1.5 r) wave-instructions (PMC: 5,000 per span and line list of the
        // 100-2500 cm^-1 cell, 13,800 per span of its merged job), every further wave sharing it repeats ~600 of them
        // (series reduction, polynomial, start-up, output), and a SIMD holding w < 4 waves runs at 0.51 / 0.82 / 0.96 of its
        // 4-wave rate (measured occupancy curve).  It reproduces the measured choices: the whole cell unsplit
        // (per-list and merged; the old line-count rule split the merged job in two: 0.288 vs 0.263 ms), the 500-900 cm^-1
        // cell in two, a merged shard of 8 in four (0.060 vs 0.072 in two, 0.112 unsplit).
        const long long want_waves = 8 * cus;
        auto simd_rate = [](double w) {
            static const double pt[5] = {0.0, 0.507, 0.818, 0.958, 1.0};
            if (w >= 4.0) return 1.0;
            const int i = (int)w;
            return pt[i] + (pt[i + 1] - pt[i]) * (w - i);
        };
        int best_R = 1, best_LS = 1;
        bool found = false;
        for (int r = 4; r >= 1 && !found; r >>= 1) {
            if (r > 1 && 64LL * r > 2 * min_H + 1) continue;
            const long long spans = std::max<long long>((total_points + 64LL * r - 1) / (64LL * r), 1);
            const double lps = lines_per_span(r);
            int cap = 1;                                   // at least one 64-line chunk per wave of a span
            while (cap < 8 && lps >= 128.0 * cap) cap <<= 1;
            const double W = 600.0 + lps * (1.5 + 1.5 * r);
            double best_cost = 0.0;
            int ls_pick = 1;
            // (measured and dropped: a three-wave workgroup per span, so that a merged shard of 8 fits the chip's 4,096 wave
            //  slots - 0.0686 ms against 0.0589 with four waves, whose second, partial round costs less than the longer chains)
            for (int ls = 1; ls <= cap; ls <<= 1) {
                const double cost = (double)spans * (W + (ls - 1) * 600.0) / std::max(simd_rate((double)spans * ls / (4.0 * cus)), 1e-3);
                if (ls == 1 || cost < 0.97 * best_cost) { best_cost = cost; ls_pick = ls; }
            }
            best_R = r; best_LS = ls_pick;
            found = spans * best_LS >= want_waves || r == 1;
        }
        *R_out = best_R; *LS_out = best_LS;
        return;
    }
    if (!R) {
        if (variant == 4) {
            R = 4;                       // work is split by lines, not spans: no reason to shrink R on small grids
        } else if (lds) {
            R = 4;
            // positional order only: finer spans shorten the tail on small grids; the longest-first
            // schedule removes that tail and then the lower instruction count of R = 4 wins
            while (!ctx->lpt && R > 2 && total_points / (64LL * R) < 8 * cus) R >>= 1;
        } else {
            R = 8;
            while (R > 1 && total_points / (64LL * R) < 12 * cus) R >>= 1;
        }
        while (R > 1 && 64LL * R > 2 * min_H + 1) R >>= 1;
    }
    if (!LS) {
        const double lps = lines_per_span(R);
        if (variant == 5)         // far lines are ~40x cheaper: a span carries less work, split it less
            LS = lps >= 4096.0 ? 4 : lps >= 1024.0 ? 2 : 1;
        else
            LS = lps >= 1024.0 ? 4 : lps >= 256.0 ? 2 : 1;
    }
    *R_out = R; *LS_out = LS;
}

// Longest-first schedule of one launch group.  Workgroups differ 7x in length (line density), and
// in positional order a dense region that happens to start late is the kernel's tail (C2: half of
// the CUs idle for the last third of the kernel).  The host knows every line's centre index (same
// IEEE expression as K1) and counts the lines each tile will walk; the sorted (job, tile) list
// depends only on the line lists and the grid, so it is built once and reused across calls
// (temperature, pressure-independent).  The dispatch order never changes a result.
//
// The same pass tabulates the line ranges of every span of 64*R points (what wave_line_ranges[_far]
// would search for): 6 lower bounds per span.  A wave then starts with one 32-byte load instead of
// six dependent probes of the centre-index array, which was most of a wave's lifetime on narrow
// windows (upper layers of a column: ~10 lines per span).  These ranges DO decide which lines a
// span visits: they are the lower bounds of the very centre indices K1 writes, because the host
// evaluates the same correctly rounded IEEE expression (nu - range_min) / resolution on the same
// doubles (host and device code are built without fast-math), and truncation is shared.
#ifndef LBL_COST_GAUSS
#define LBL_COST_GAUSS 15.0      // wave-instructions a near line's Gaussian passes add per span (0.6 passes of ~25; 29 before the 16-point runs)
#endif
// list_first (merged layer jobs, else NULL): job j accumulates the line lists [list_first[j], list_first[j + 1]) of `lines`
// as ONE record array; without it job j is line list j.
static lbl_ctx::Schedule* group_schedule(lbl_ctx* ctx, int variant, const std::vector<int>& jobs_in_group,
                                         lbl_lines* const* lines, const lbl_grid* grid, int R, int LS, long long tile_pts,
                                         const int* list_first = nullptr) {
    std::vector<uint64_t> key;
    const bool far_field = variant == 5;
    int far_half_spans = 0;
    double far_cost = 1.0;
    if (far_field) accumulate_far_field_params(R, &far_half_spans, &far_cost, ctx->accuracy);      // (the span tables hold the far bounds)
    auto l0 = [&](int j) { return list_first ? list_first[j] : j; };
    auto l1 = [&](int j) { return list_first ? list_first[j + 1] : j + 1; };
    bool merged = false;                       // any job of the group with several line lists
    for (int j : jobs_in_group) merged = merged || l1(j) - l0(j) > 1;
    // (a merged group's tables and merged positions only come from the device build, whatever the options say)
    const int build_bits = merged ? 1 : ctx->sched_build, lpt_bits = merged ? 4 : ctx->lpt;
    // XCD-partitioned order: contiguous chunks of the tile sequence per XCD.  Every chunk drags the line halo of its ends into
    // its XCD's L2, so few chunks mean little traffic, and many chunks even out what the cost model misjudges by spectral
    // region: about 29 workgroups per chunk, between 10 and 32 chunks per XCD (round 2 measured 32 against 8 and 1 on the
    // per-list 100-2500 cm^-1 cell, 7,031 workgroups; round 5 on its merged job, 2,344 workgroups: 10 chunks fetch 34 MB
    // where 32 fetch 55, same kernel time; a merged shard of 8, 1,172 workgroups: 10.7 MB against 16.5, same time; 6: +8 %)
    long long n_wg = 0;
    for (int j : jobs_in_group) { long long sf, sc; shard_range(grid[j], &sf, &sc); n_wg += (sc + tile_pts - 1) / tile_pts; }
    const int xcd_chunks = ctx->xcd_chunks > 0 ? ctx->xcd_chunks : (int)std::min<long long>(32, std::max<long long>(10, (n_wg + 116) / 232));
    // (a launch of one round: 1..16 runs per XCD, default 1 - see launch_schedule_build)
    const int sr_chunks = ctx->xcd_chunks > 0 ? std::min(ctx->xcd_chunks, 16) : 1;
    key.push_back((uint64_t)(ctx->xcd_tol + 1) << 59 | (uint64_t)sr_chunks << 52 | (uint64_t)ctx->xcd_pack << 48 | (uint64_t)xcd_chunks << 40 | (uint64_t)R << 32 | (uint64_t)far_half_spans << 16 | (uint64_t)LS << 8 | (uint64_t)build_bits << 4 | (uint64_t)lpt_bits << 1 | (uint64_t)far_field);
    for (int j : jobs_in_group) {
        long long sf, sc;
        shard_range(grid[j], &sf, &sc);
        uint64_t bits_a, bits_b;
        memcpy(&bits_a, &grid[j].range_min, 8); memcpy(&bits_b, &grid[j].resolution, 8);
        if (list_first) key.push_back(0xFFFFFFFF00000000ull | (uint64_t)(l1(j) - l0(j)));      // (a job of k lists is not k jobs)
        for (int l = l0(j); l < l1(j); ++l) { key.push_back(lines[l]->serial); key.push_back((uint64_t)lines[l]->n); }
        key.push_back(bits_a); key.push_back(bits_b);
        key.push_back((uint64_t)sf); key.push_back((uint64_t)sc); key.push_back((uint64_t)grid[j].window);
    }
    for (size_t i = 0; i < ctx->schedules.size(); ++i)
        if (ctx->schedules[i]->key == key) {
            // least recently used at the front: an entry handed out in this call is never the next to go
            std::rotate(ctx->schedules.begin() + i, ctx->schedules.begin() + i + 1, ctx->schedules.end());
            return ctx->schedules.back().get();
        }
    if (ctx->capturing) { (void)capture_refuses(ctx, "building a dispatch schedule"); return nullptr; }
    TraceScope tr("group_schedule", (long long)jobs_in_group.size());
    auto evict_oldest = [&]() {
        // (64 entries: a 30-layer atmosphere used through per-layer getters holds one schedule per layer, and every eviction is a
        // stream sync, a hipFree and a stale graph - advisor, round 5; an entry is a few MB: dispatch list, span table, merged order)
        if (ctx->schedules.size() >= 64) {                    // drop the least recently used entry
            ctx->epoch++;
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->schedules.front()->d_block) (void)hipFree(ctx->schedules.front()->d_block);
            ctx->schedules.erase(ctx->schedules.begin());
        }
    };
    auto one_block = [&](size_t n_items, size_t n_tab_ints, int2** d_list, int32_t** d_tabs, size_t n_src = 0,
                         int32_t** d_src = nullptr) -> void* {
        const size_t list_bytes = (std::max<size_t>(n_items, 1) * sizeof(int2) + 255) & ~(size_t)255;
        const size_t tab_bytes = (std::max<size_t>(n_tab_ints, 8) * sizeof(int32_t) + 255) & ~(size_t)255;
        void* blk = nullptr;
        TraceScope tr("schedule block alloc", (long long)(list_bytes + tab_bytes + n_src * sizeof(int32_t)));
        if (hipMalloc(&blk, list_bytes + tab_bytes + n_src * sizeof(int32_t)) != hipSuccess) return nullptr;
        *d_list = (int2*)blk;
        *d_tabs = (int32_t*)((char*)blk + list_bytes);
        if (d_src) *d_src = n_src ? (int32_t*)((char*)blk + list_bytes + tab_bytes) : nullptr;
        return blk;
    };
    {
        // Device build: only the sizes are worked out here; the tables and the order are produced in stream by the first
        // batch that uses the schedule (enqueue_accumulate, after its line prep has written the centre indices).
        long long n_items = 0, n_spans_all = 0;
        for (int j : jobs_in_group) {
            long long sf, sc;
            shard_range(grid[j], &sf, &sc);
            n_items += (sc + tile_pts - 1) / tile_pts;
            n_spans_all += (sc + 64LL * R - 1) / (64LL * R);
        }
        const int n_cu_i = ctx->n_cu > 0 ? ctx->n_cu : 256;
        const bool device_ok = n_items > 0 && n_spans_all < (1LL << 27) && sched_device_supported((int)n_items, n_cu_i);
        if (merged && !device_ok) { ctx->sched_refused = true; return nullptr; }
        if (build_bits == 1 && lpt_bits == 4 && device_ok) {
            std::unique_ptr<lbl_ctx::Schedule> S(new lbl_ctx::Schedule());
            S->key = key;
            S->R = R; S->spans_per_tile = (int)(tile_pts / (64LL * R));
            S->far_reach = far_field ? (long long)far_half_spans * 32 * R : 0;
            S->xcd_chunks = xcd_chunks;
            S->cost_near = 5.0 * R + LBL_COST_GAUSS; S->cost_edge = 8.0 * R; S->cost_far = far_cost * 5.0 * R; S->cost_fixed = 600.0;
            int32_t span_run = 0, tile_run = 0;
            for (int j : jobs_in_group) {
                long long sf, sc;
                shard_range(grid[j], &sf, &sc);
                S->tab_off.push_back((size_t)span_run * 8);
                S->span_first.push_back(span_run); S->tile_first.push_back(tile_run);
                span_run += (int32_t)((sc + 64LL * R - 1) / (64LL * R));
                tile_run += (int32_t)((sc + tile_pts - 1) / tile_pts);
            }
            S->total_spans = span_run; S->total = tile_run;
            S->xcd_pack = ctx->xcd_pack; S->sr_chunks = sr_chunks; S->xcd_tol = ctx->xcd_tol;
            S->launch_total = sched_launch_items(tile_run, n_cu_i, S->xcd_pack != 0);
            size_t n_src = 0;
            for (int j : jobs_in_group) {
                size_t n = 0;
                for (int l = l0(j); l < l1(j); ++l) n += (size_t)lines[l]->n;
                S->src_off.push_back(l1(j) - l0(j) > 1 ? n_src : SIZE_MAX);
                if (l1(j) - l0(j) > 1) n_src += n;
            }
            S->d_block = one_block((size_t)S->launch_total, (size_t)span_run * 8, &S->d_list, &S->d_tabs, n_src, &S->d_src);
            if (!S->d_block) return nullptr;
            S->pending = true;
            S->merge_pending = n_src > 0;
            evict_oldest();
            ctx->schedules.push_back(std::move(S));
            return ctx->schedules.back().get();
        }
    }
    struct Item { int count, job, tile; };
    std::vector<Item> items;
    std::vector<long long> idx;
    std::vector<int32_t> tabs;
    std::vector<size_t> tab_off;
    const long long span = 64LL * R;
    const long long reach = (long long)far_half_spans * 32 * R;
    for (size_t k = 0; k < jobs_in_group.size(); ++k) {
        const int j = jobs_in_group[k];
        const lbl_lines* L = lines[l0(j)];           // (one list per job on this path: merged groups took the device build)
        idx.resize((size_t)L->n);
        for (int64_t i = 0; i < L->n; ++i) idx[i] = (long long)((L->host_nu[i] - grid[j].range_min) / grid[j].resolution);
        long long sf, sc;
        shard_range(grid[j], &sf, &sc);
        const long long H = std::max<long long>(grid[j].window - 2, 0);
        auto below = [&](long long v) { return (int32_t)(std::lower_bound(idx.begin(), idx.end(), v) - idx.begin()); };
        // span table (same arithmetic as wave_line_ranges_far)
        const long long n_spans = (sc + span - 1) / span;
        tab_off.push_back(tabs.size());
        tabs.resize(tabs.size() + (size_t)n_spans * 8, 0);
        int32_t* T = tabs.data() + tab_off.back();
        for (long long t = 0; t < n_spans; ++t) {
            const long long lo = sf + t * span, hi = std::min(lo + span - 1, sf + sc - 1);
            int32_t iA = below(lo - H), iB = below(hi - H), iC = below(lo + H + 1), iD = below(hi + H + 1);
            if (hi - H >= lo + H + 1) { iB = iD; iC = iD; }
            int32_t iF1 = iB, iF2 = iC;
            if (far_field) {
                iF1 = std::min(std::max(below(lo + 32 * R - reach), iB), iC);      // first line with c > fl
                iF2 = std::min(std::max(below(lo + 32 * R + reach), iF1), iC);     // first line with c >= fr
            }
            int32_t iN1 = iF1, iN2 = iF2;                                          // the bounds at 3 half-spans (FF_MID), inside [iF1, iF2]
            if (far_field) {
                iN1 = std::min(std::max(below(lo + 32 * R - 3LL * 32 * R), iF1), iF2);
                iN2 = std::min(std::max(below(lo + 32 * R + 3LL * 32 * R), iN1), iF2);
            }
            int32_t* e = T + t * 8;
            e[0] = iA; e[1] = iB; e[2] = iC; e[3] = iD; e[4] = iF1; e[5] = iF2; e[6] = iN1; e[7] = iN2;
        }
        // cost of every tile (a workgroup's points): lines it walks, far lines at their series price
        const long long n_tiles = (sc + tile_pts - 1) / tile_pts;
        const long long spans_per_tile = tile_pts / span;
        for (long long t = 0; t < n_tiles; ++t) {
            double cost = 0.0;
            for (long long q = t * spans_per_tile; q < std::min((t + 1) * spans_per_tile, n_spans); ++q) {
                // wave-instructions of one span: near line 5R + ~0.6 Gaussian passes (LBL_COST_GAUSS), masked edge
                // line 8R, series line ~1.6, fixed part ~600 (variants without the series: all direct)
                const int32_t* e = T + q * 8;
                const double n_far = (double)((e[4] - e[1]) + (e[2] - e[5]));
                const double n_edge = (double)((e[1] - e[0]) + (e[3] - e[2]));
                const double n_near = (double)(e[5] - e[4]);
                cost += n_near * (5.0 * R + LBL_COST_GAUSS) + n_edge * 8.0 * R + n_far * far_cost * 5.0 * R + 600.0;
            }
            items.push_back({(int)(cost + 0.5), (int)k, (int)t});
        }
    }
    const size_t n_cu = (size_t)(ctx->n_cu > 0 ? ctx->n_cu : 256);
    bool xcd_done = false;
    if (ctx->lpt == 4 && items.size() > 4 * n_cu) {
        // XCD-partitioned longest-first (launches of several rounds): workgroup i is dispatched to XCD
        // i mod 8, and every XCD has its own L2.  The positional sequence of tiles is cut into 8
        // contiguous parts of equal estimated cost, each part sorted longest-first, and the parts are
        // interleaved, so XCD x only ever reads the line records of part x (an eighth of the list plus
        // the window halo) instead of every XCD pulling every record into its L2.
        const int X = 8;
        // (much finer than one part per XCD: 32 chunks per XCD, dealt round-robin, so that every XCD samples
        // the whole spectrum - a bias of the cost estimate in one spectral region, e.g. the cheaper
        // pure-Lorentz lines at low wavenumbers, then loads all XCDs alike.  Measured on the 100-2500 cm^-1
        // cell: 1 chunk per XCD +6 % kernel time, 8 chunks +3 %, 32 chunks +-0 with 33 MB fetched instead of 110)
        const int chunks = X * xcd_chunks;
        double total = 0.0;
        for (const Item& it : items) total += it.count;
        std::vector<std::vector<Item>> part(X);
        double run = 0.0;
        for (const Item& it : items) {
            int c = (int)(run / (total / chunks + 1e-9));
            if (c >= chunks) c = chunks - 1;
            part[(size_t)(c % X)].push_back(it);
            run += it.count;
        }
        for (auto& p : part) std::stable_sort(p.begin(), p.end(), [](const Item& a, const Item& b) { return a.count > b.count; });
        std::vector<Item> out;
        out.reserve(items.size());
        std::vector<size_t> pos(X, 0);
        while (out.size() < items.size()) {
            for (int x = 0; x < X; ++x) {
                int src = x;
                if (pos[(size_t)src] >= part[(size_t)src].size()) {       // part x is exhausted: its slots take from the part with most left
                    size_t best_left = 0;
                    src = -1;
                    for (int y = 0; y < X; ++y) {
                        const size_t left = part[(size_t)y].size() - pos[(size_t)y];
                        if (left > best_left) { best_left = left; src = y; }
                    }
                    if (src < 0) break;
                }
                out.push_back(part[(size_t)src][pos[(size_t)src]++]);
            }
        }
        items.swap(out);
        xcd_done = true;
    }
    if (!xcd_done)
        std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.count > y.count; });
    if (xcd_done) {
    } else if (ctx->lpt == 2) {
        // snake order: on a small grid every workgroup is resident from the first cycle, nothing is
        // dispatched dynamically, and CU k receives items k, k + n_cu, k + 2 n_cu, ...: reversing
        // every other tier of n_cu items pairs the heaviest of one tier with the lightest of the next
        for (size_t b = n_cu; b < items.size(); b += 2 * n_cu)
            std::reverse(items.begin() + b, items.begin() + std::min(items.size(), b + n_cu));
    } else if (ctx->lpt >= 3 && items.size() <= 4 * n_cu) {
        // single round: pack the items into n_cu bins longest-first (each to the least loaded bin
        // that still has a free slot), then emit bin-interleaved so that the dispatcher's round
        // robin over the CUs rebuilds the bins
        const size_t slots = (items.size() + n_cu - 1) / n_cu;
        std::vector<std::vector<Item>> bins(n_cu);
        std::vector<long long> load(n_cu, 0);
        for (const Item& it : items) {
            size_t best = n_cu;
            for (size_t b = 0; b < n_cu; ++b)
                if (bins[b].size() < slots && (best == n_cu || load[b] < load[best])) best = b;
            bins[best].push_back(it);
            load[best] += it.count;
        }
        std::vector<Item> out;
        out.reserve(items.size());
        for (size_t t = 0; t < slots; ++t)
            for (size_t b = 0; b < n_cu; ++b)
                if (t < bins[b].size()) out.push_back(bins[b][t]);
        items.swap(out);
    }
    std::vector<int2> host(items.size());
    for (size_t i = 0; i < items.size(); ++i) { host[i].x = items[i].job; host[i].y = items[i].tile; }
    int2* d_list = nullptr;
    int32_t* d_tabs = nullptr;
    void* blk = one_block(host.size(), tabs.size(), &d_list, &d_tabs);
    if (!blk) return nullptr;
    if ((!host.empty() && hipMemcpy(d_list, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess) ||
        (!tabs.empty() && hipMemcpy(d_tabs, tabs.data(), tabs.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipFree(blk);
        return nullptr;
    }
    evict_oldest();
    std::unique_ptr<lbl_ctx::Schedule> S(new lbl_ctx::Schedule());
    S->key = key; S->d_list = d_list; S->total = (int)host.size(); S->launch_total = S->total; S->d_tabs = d_tabs; S->tab_off = tab_off; S->d_block = blk;
    S->total_spans = -(int)(tabs.size() / 8);          // (negative: built on the host; lbl_schedule_export)
    ctx->schedules.push_back(std::move(S));
    return ctx->schedules.back().get();
}

// Merged layer jobs (lbl_layer_merged_step_dev, lbl_layers_merged_accumulate_dev): accumulate job j takes the line lists
// [first[j], first[j + 1]) of `lines` / `iso` as ONE record array in centre-index order; list l's amplitudes carry weight[l]
// and job j's sums leave the kernel multiplied by out_scale[j] (see PrepJob.weight, AccumJob.out_scale).
struct MergeSpec {
    const int* first;          // n_jobs + 1 offsets into lines / iso
    const double* weight;      // per line list
    const double* out_scale;   // per job
};

static int enqueue_accumulate(lbl_ctx* ctx, int n_jobs, lbl_lines* const* lines, const lbl_iso_params* iso,
                              const lbl_grid* grid, double* const* out_dev, bool prep_only,
                              const FusedSweep* fuse = nullptr, double fuse_conc = 0.0, const MergeSpec* merge = nullptr) {
    // fuse != NULL (n_jobs == 1): the layer sweep runs in the output stage of the one job
    // merge == NULL: job j is line list j; lines / iso hold n_jobs entries.  grid and out_dev are per JOB either way.
    if (n_jobs <= 0) return LBL_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto l0 = [&](int j) { return merge ? merge->first[j] : j; };
    auto l1 = [&](int j) { return merge ? merge->first[j + 1] : j + 1; };
    const int n_lists = l1(n_jobs - 1);
    if (merge && !(ctx->accum_variant == 3 || ctx->accum_variant == 5))
        return fail(ctx, LBL_ERR_BAD_ARG, "merged layer jobs run on the LDS accumulate kernels only (accum_variant 3 or 5)");
    // layout of the scratch arenas: the lists of a job are neighbours, so a job's merged record array is one range
    std::vector<size_t> line_off(n_lists), work_off(n_jobs), job_lines(n_jobs, 0);
    size_t tot_lines = 0, tot_work = 0;
    long long min_H = 1LL << 40;
    int max_lines = 0;
    for (int j = 0; j < n_jobs; ++j) {
        int rc = check_grid(ctx, &grid[j]);
        if (rc) return rc;
        for (int l = l0(j); l < l1(j); ++l) {
            if (!lines[l] || lines[l]->ctx != ctx) return fail(ctx, LBL_ERR_STATE, "line list %d: missing or from another context", l);
            if (!(iso[l].T > 0) || !(iso[l].P > 0) || !(iso[l].molmass > 0) || !(iso[l].Q_T > 0))
                return fail(ctx, LBL_ERR_BAD_ARG, "line list %d: T, P, molmass and Q_T must be > 0", l);
            line_off[l] = tot_lines;
            tot_lines += (size_t)lines[l]->n;
            job_lines[j] += (size_t)lines[l]->n;
            max_lines = std::max<int>(max_lines, (int)lines[l]->n);
        }
        if (job_lines[j] > 2000000000ull) return fail(ctx, LBL_ERR_BAD_ARG, "job %d: too many lines for int32 indexing", j);
        if (l1(j) - l0(j) > 1) {                 // merged-order map: list within the job in 6 bits, line within the list in 26
            if (l1(j) - l0(j) > kMaxIso) return fail(ctx, LBL_ERR_BAD_ARG, "job %d: at most %d line lists per merged job", j, kMaxIso);
            for (int l = l0(j); l < l1(j); ++l)
                if (lines[l]->n >= (1LL << 26)) return fail(ctx, LBL_ERR_BAD_ARG, "line list %d: at most 2^26 - 1 lines per list of a merged job", l);
        }
        work_off[j] = tot_work;
        if (needs_regrid(grid[j])) tot_work += (size_t)grid[j].n_work;
        min_H = std::min<long long>(min_H, std::max<long long>(grid[j].window - 2, 0));
    }
    int rc;
    if ((rc = arena_reserve(ctx, ctx->recs, std::max<size_t>(tot_lines, 1) * sizeof(HotRec)))) return rc;
    if ((rc = arena_reserve(ctx, ctx->cold, std::max<size_t>(tot_lines, 1) * sizeof(ColdRec)))) return rc;
    if ((rc = arena_reserve(ctx, ctx->cidx, std::max<size_t>(tot_lines, 1) * sizeof(int32_t)))) return rc;
    if ((rc = arena_reserve(ctx, ctx->work, std::max<size_t>(tot_work, 1) * sizeof(double)))) return rc;
    // per-block regime counts (3 x u32 per block of 256 lines), summed on the host on demand
    // (a list of a merged job is counted per block of 256 MERGED positions of its job)
    int n_merged_jobs = 0;
    size_t max_job_lines = 0;
    for (int j = 0; j < n_jobs; ++j)
        if (l1(j) - l0(j) > 1) { ++n_merged_jobs; max_job_lines = std::max(max_job_lines, job_lines[j]); }
    const int blocks_per_job = (int)((std::max<size_t>((size_t)max_lines, max_job_lines) + 255) / 256);
    const size_t prep_bytes = (((size_t)n_lists * sizeof(PrepJob) + (size_t)n_merged_jobs * sizeof(MergedPrep)) + 15) & ~(size_t)15;
    const size_t acc_bytes = (size_t)n_jobs * sizeof(AccumJob);
    const size_t cnt_bytes = (size_t)n_lists * std::max(blocks_per_job, 1) * 3 * sizeof(unsigned int);
    if ((rc = arena_reserve(ctx, ctx->counts, cnt_bytes))) return rc;
    // descriptors are built in pageable host memory first: a batch that repeats (the usual case:
    // same layer, step after step) finds its descriptor block already on the device and skips the copy
    ctx->desc_build.assign(prep_bytes + acc_bytes, 0);
    void* stage = ctx->desc_build.data();
    unsigned int* d_counts = (unsigned int*)ctx->counts.ptr;

    // Jobs of one batch can have very different windows (a column: W = 5000 at the surface,
    // 50 at 10 mbar).  A wave must not own more points than a line's support is wide, so jobs are
    // grouped by the largest R their window allows and every group gets its own launch shape.
    const int r_default = ctx->accum_variant >= 3 ? 4 : 8;       // what choose_shape starts from
    auto r_cap = [&](int j) {
        const long long H = std::max<long long>(grid[j].window - 2, 0);
        int r = r_default;                        // windows that allow more than the default share its group
        while (r > 1 && 64LL * r > 2 * H + 1) r >>= 1;
        return ctx->accum_R ? 8 : r;              // a forced R keeps one group
    };
    // Far-field kernel: a window narrower than (threshold + 1) half-spans has no far line on any span;
    // such jobs run the all-direct kernel (same arithmetic there, 60-90 instead of 105-123 VGPRs, so
    // 5-8 instead of 4 waves per SIMD: narrow-window spans are latency bound).
    int far_half_spans = 0;
    { double fc; accumulate_far_field_params(4, &far_half_spans, &fc); }
    auto has_far = [&](int j) {
        if (ctx->accum_variant != 5) return 0;
        const long long H = std::max<long long>(grid[j].window - 2, 0);
        const long long r = ctx->accum_R ? ctx->accum_R : r_cap(j);
        if (ctx->skew && H < ctx->far_min_H) return 0;
        return H >= 32 * r * (far_half_spans + 1) ? 1 : 0;
    };
    // Narrow windows (no far line on any span) go to the skewed-range kernel, all in ONE group whatever their
    // width (its lanes walk per-lane line ranges, so a span may be wider than a line's support), provided the
    // group fills the chip: at least 8 wavefronts per CU, else the span kernel's line split serves small grids better.
    bool skew_on = false;
    if (ctx->accum_variant == 5 && ctx->skew && !ctx->accum_R && !ctx->accum_LS) {
        long long pts = 0;
        for (int j = 0; j < n_jobs; ++j)
            if (ctx->skew == 2 || !has_far(j)) { long long f, c; shard_range(grid[j], &f, &c); pts += c; }
        const long long cus = ctx->n_cu > 0 ? ctx->n_cu : 256;
        skew_on = ctx->skew == 2 || pts / (64LL * ctx->skew_R) >= 8 * cus;
    }
    auto is_skew = [&](int j) { return skew_on && (ctx->skew == 2 || !has_far(j)); };
    auto group_key = [&](int j) { return is_skew(j) ? 0 : r_cap(j) * 2 + has_far(j); };
    std::vector<int> order(n_jobs);
    for (int j = 0; j < n_jobs; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return group_key(a) > group_key(b); });
    struct Group { int first, count, R, LS, max_tiles, variant; int grun; const int2* worklist; int total_tiles; const int32_t* tabs; std::vector<size_t> tab_off; lbl_ctx::Schedule* sched; };
    std::vector<Group> groups;
    for (int k = 0; k < n_jobs;) {
        int e = k;
        long long pts = 0, lns = 0, mh = 1LL << 40, mxh = 0;
        while (e < n_jobs && group_key(order[e]) == group_key(order[k])) {
            long long f, c;
            shard_range(grid[order[e]], &f, &c);
            pts += c; lns += (long long)job_lines[order[e]];
            const long long H = std::max<long long>(grid[order[e]].window - 2, 0);
            mh = std::min(mh, H); mxh = std::max(mxh, H);
            ++e;
        }
        Group g{k, e - k, 0, 0, 0, ctx->accum_variant, 16, nullptr, 0, nullptr, {}, nullptr};
        if (is_skew(order[k])) {
            g.R = ctx->skew_R; g.LS = 1; g.variant = 6;
            // dense (merged) line lists: a chunk of 80 records must cover about the span's 64 R points, else only part of
            // the lanes has lines in it - the waves of a workgroup then share a span and deal its records (R = 8 only)
            const double chunk_cover = pts > 0 ? 80.0 * (double)pts / std::max<double>((double)lns, 1.0) / (64.0 * g.R) : 1e9;
            if (g.R == 8) g.LS = ctx->skew_LS ? ctx->skew_LS : (chunk_cover < 1.5 ? 4 : chunk_cover < 3.0 ? 2 : 1);
        } else {
            choose_shape(ctx, g.variant, pts, lns, mh, &g.R, &g.LS);
            // with the R actually chosen (a small grid may have shrunk it): does any job of the group have far lines?
            if (ctx->accum_variant == 5 && mxh < 32LL * g.R * (far_half_spans + 1)) g.variant = 3;
            // Gaussian runs of 32 points (the far-field kernel's three-waves-per-SIMD build, lbl_kernels.hip).  Budget mode: for
            // launches of more than one round of the chip's wave slots at four per SIMD, a launch that fits one round keeps the
            // 16-point build.  Exact mode: the 32-point build is also the one whose series starts at 3 half-spans (fewer
            // instructions per span), so it takes every launch except those that fit one round at four waves per SIMD but not
            // at three (column of 50 layers, forced 16 / this rule / forced 32, same box: 3.797 / 3.623 -> 3.60 / 3.604 ms).
            if (g.variant == 5 && g.R == 4 && g.LS == 1) {
                const long long waves = (pts + 255) / 256, n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
                const bool run32 = ctx->accuracy ? waves > 16 * n_cu : !(waves > 12 * n_cu && waves <= 16 * n_cu);
                g.grun = ctx->gauss_run ? ctx->gauss_run : (run32 ? 32 : 16);
            }
        }
        if ((g.variant == 3 || g.variant == 5 || g.variant == 6) && (ctx->lpt || merge)) {
            // cached schedule of this group: dispatch order + the line ranges of every span (+ merged positions)
            std::vector<int> members(order.begin() + k, order.begin() + e);
            ctx->sched_refused = false;
            lbl_ctx::Schedule* sc = group_schedule(ctx, g.variant == 6 ? 3 : g.variant, members, lines, grid, g.R, g.LS,
                                                   accumulate_tile_points(g.R, g.LS, g.variant), merge ? merge->first : nullptr);
            if (!sc && ctx->sched_refused)
                return fail(ctx, LBL_ERR_BAD_ARG, "merged layer jobs need the device schedule build, which does not cover a launch of this size: use the per-line-list step");
            if (!sc) return ctx->capturing ? LBL_ERR_STATE : fail(ctx, LBL_ERR_OOM, "schedule allocation failed");
            if ((sc->pending || sc->merge_pending) && ctx->capturing) return capture_refuses(ctx, "building a dispatch schedule");
            g.worklist = sc->d_list; g.total_tiles = sc->launch_total; g.tabs = sc->d_tabs; g.tab_off = sc->tab_off; g.sched = sc;
        }
        groups.push_back(g);
        k = e;
    }
    // where the group's schedule keeps the merged-order map of job order[k] (NULL: a job of one list)
    std::vector<const int32_t*> job_src(n_jobs, nullptr);
    for (const Group& g : groups)
        for (int k = g.first; k < g.first + g.count; ++k)
            if (g.sched && g.sched->d_src && g.sched->src_off[(size_t)(k - g.first)] != SIZE_MAX)
                job_src[order[k]] = g.sched->d_src + g.sched->src_off[(size_t)(k - g.first)];
    PrepJob* hp = (PrepJob*)stage;
    MergedPrep* hm = (MergedPrep*)((char*)stage + (size_t)n_lists * sizeof(PrepJob));
    AccumJob* ha = (AccumJob*)((char*)stage + prep_bytes);
    ctx->last_list_blocks.assign(n_lists, 0);
    int mj = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const size_t base = line_off[l0(j)];                          // the job's record array
        if (job_src[j]) {
            MergedPrep& m = hm[mj++];
            memset(&m, 0, sizeof m);
            m.src = job_src[j];
            m.hot = (HotRec*)ctx->recs.ptr + base; m.cold = (ColdRec*)ctx->cold.ptr + base; m.cidx = (int32_t*)ctx->cidx.ptr + base;
            m.first_list = l0(j); m.n_lists = l1(j) - l0(j);
            m.n_total = (int32_t)job_lines[j]; m.blocks = (int32_t)((job_lines[j] + 255) / 256);
        }
        for (int l = l0(j); l < l1(j); ++l) {
            ctx->last_list_blocks[l] = job_src[j] ? (int)((job_lines[j] + 255) / 256) : (int)((lines[l]->n + 255) / 256);
            const lbl_lines* L = lines[l];
            PrepJob& p = hp[l];
            memset(&p, 0, sizeof p);
            p.nu = L->field(0); p.sw = L->field(1); p.elower = L->field(2); p.gamma_air = L->field(3);
            p.gamma_self = L->field(4); p.n_air = L->field(5); p.delta_air = L->field(6);
            const bool scattered = job_src[j] != nullptr;            // several lists: prepared in merged order into the job's arrays
            const size_t off = scattered ? base : line_off[l];
            p.hot = (HotRec*)ctx->recs.ptr + off;
            p.cold = (ColdRec*)ctx->cold.ptr + off;
            p.cidx = (int32_t*)ctx->cidx.ptr + off;
            p.merged = scattered ? 1 : 0;
            p.weight = merge ? merge->weight[l] : 1.0;
            p.block_counts = d_counts + (size_t)l * blocks_per_job * 3;
            p.T = iso[l].T; p.P = iso[l].P; p.q_frac = iso[l].q_frac; p.molmass = iso[l].molmass;
            p.Q_T = iso[l].Q_T; p.Q_296 = iso[l].Q_296;
            p.range_min = grid[j].range_min; p.resolution = grid[j].resolution;
            p.log_t0_over_T = std::log(296.0 / iso[l].T);
            p.P_over_p0 = iso[l].P / p0;
            { const double m = iso[l].molmass / 1000.0 / avo; p.ghw_factor = std::sqrt(2.0 * kB * iso[l].T / m / (cLight * cLight)); }
            p.q_ratio = iso[l].Q_296 / iso[l].Q_T;
            p.inv_T = 1.0 / iso[l].T; p.inv_res = 1.0 / grid[j].resolution; p.inv_res2 = p.inv_res * p.inv_res;
            p.gauss_cut = ctx->accuracy ? 17179869184.0 : 18014398509481984.0;          // 2^34 : 2^54
            p.n_lines = (int32_t)L->n;
            p.pad = ctx->ablate;                 // (LBL_DIAG builds: debug_ablate 1024 drops every Gaussian part; else 0, never read)
        }
    }
    const bool balanced = ctx->accum_variant == 4;
    // balanced variant scratch, per group: SpanRec[S] | counts u32[S] | prefix u64[S+1] | slab
    std::vector<size_t> bal_off(groups.size() + 1, 0);
    std::vector<int> group_spans(groups.size(), 0), group_workers(groups.size(), 0);
    for (Group& g : groups) {
        const long long tile_pts = accumulate_tile_points(g.R, g.LS, g.variant);
        for (int k = g.first; k < g.first + g.count; ++k) {
            const int j = order[k];
            AccumJob& a = ha[k];
            memset(&a, 0, sizeof a);
            a.hot = (const HotRec*)ctx->recs.ptr + line_off[l0(j)];
            a.cold = (const ColdRec*)ctx->cold.ptr + line_off[l0(j)];
            a.cidx = (const int32_t*)ctx->cidx.ptr + line_off[l0(j)];
            a.out = needs_regrid(grid[j]) ? (double*)ctx->work.ptr + work_off[j] : out_dev[j];
            a.n_lines = (int32_t)job_lines[j];
            a.n_work = (int32_t)grid[j].n_work;
            a.H = (int32_t)std::max<long long>(grid[j].window - 2, 0);
            long long sf, sc;
            shard_range(grid[j], &sf, &sc);
            a.p_begin = (int32_t)sf;
            a.p_end = (int32_t)(sf + sc);
            a.n_tiles = (int32_t)((sc + tile_pts - 1) / tile_pts);
            a.flush_every = (a.H + 64 * g.R + 1 <= 40000) ? 32 : 16;
            a.pad = ctx->tile_order;
            a.ablate = ctx->ablate;
            a.span_tab = g.tabs ? g.tabs + g.tab_off[(size_t)(k - g.first)] : nullptr;
            a.out_scale = merge ? merge->out_scale[j] : 1.0;
            if (fuse && n_jobs == 1) {
                a.chain_flags = CHAIN_MOL_FIRST | CHAIN_MOL_LAST;
                a.conc = fuse_conc;
                a.fuse = *fuse;
            }
            g.max_tiles = std::max(g.max_tiles, a.n_tiles);
            if (balanced) {
                const size_t gi = (size_t)(&g - &groups[0]);
                a.span_first = group_spans[gi];
                a.n_spans = (int32_t)((sc + 64LL * g.R - 1) / (64LL * g.R));
                group_spans[gi] += a.n_spans;
            }
        }
    }
    if (balanced) {
        size_t tot = 0;
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const size_t S = (size_t)group_spans[gi];
            group_workers[gi] = ctx->bal_workers[groups[gi].R] ? ctx->bal_workers[groups[gi].R]
                                                                 : (ctx->bal_workers[groups[gi].R] = balanced_workers(groups[gi].R, ctx->n_cu));
            bal_off[gi] = tot;
            tot += ((S * sizeof(SpanRec) + S * sizeof(unsigned int) + (S + 1) * sizeof(unsigned long long) + 255) & ~(size_t)255)
                   + (size_t)group_workers[gi] * 2 * 64 * groups[gi].R * sizeof(double) + 256;
        }
        bal_off[groups.size()] = tot;
        if ((rc = arena_reserve(ctx, ctx->bal, std::max<size_t>(tot, 256)))) return rc;
    }
    // device copy of the descriptors: reuse an identical block if one is cached (4 slots, round robin)
    char* d_desc = nullptr;
    for (auto& e : ctx->desc_cache)
        if (e.dptr && e.bytes == ctx->desc_build) { d_desc = (char*)e.dptr; break; }
    if (!d_desc) {
        if (ctx->capturing) return capture_refuses(ctx, "uploading job descriptors");
        ctx->epoch++;                                      // the slot's old contents may be what a captured graph reads
        auto& e = ctx->desc_cache[ctx->desc_next];
        ctx->desc_next = (ctx->desc_next + 1) % 4;
        if (e.cap < ctx->desc_build.size()) {
            TraceScope tr("descriptor slot alloc", (long long)ctx->desc_build.size());
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (e.dptr) HIP_TRY(ctx, hipFree(e.dptr));
            e.dptr = nullptr; e.cap = 0;
            const size_t want = ctx->desc_build.size() + ctx->desc_build.size() / 4 + 1024;
            HIP_TRY(ctx, hipMalloc(&e.dptr, want));
            e.cap = want;
        }
        void* pinned = nullptr;
        if ((rc = stage_alloc(ctx, ctx->desc_build.size(), &pinned))) return rc;
        memcpy(pinned, ctx->desc_build.data(), ctx->desc_build.size());
        HIP_TRY(ctx, hipMemcpyAsync(e.dptr, pinned, ctx->desc_build.size(), hipMemcpyHostToDevice, ctx->stream));
        e.bytes = ctx->desc_build;
        d_desc = (char*)e.dptr;
    }
    PrepJob* dp = (PrepJob*)d_desc;
    AccumJob* da = (AccumJob*)(d_desc + prep_bytes);
    ctx->last_blocks_per_job = blocks_per_job;
    // Merged layer jobs whose schedule is new: the merged-order map, ahead of the line prep that runs in that order - each
    // list's centre indices (K1's expression) into scratch, then one rank search per line and other list of its job.
    // Once per (line lists, grid); kept in the schedule's block.
    for (Group& g : groups) {
        lbl_ctx::Schedule* sc = g.sched;
        if (!sc || !sc->merge_pending) continue;
        TraceScope tr("merge ranks (enqueue)", (long long)g.count);
        std::vector<MergeList> ml;
        std::vector<size_t> tmp_off;
        size_t tmp_lines = 0;
        int ml_max = 0;
        for (int k = g.first; k < g.first + g.count; ++k) {
            const int j = order[k];
            if (!job_src[j]) continue;
            const int jf = (int)ml.size();
            for (int l = l0(j); l < l1(j); ++l) {
                MergeList m;
                memset(&m, 0, sizeof m);
                m.nu = lines[l]->field(0);
                tmp_off.push_back(tmp_lines);                             // (tmp_cidx is set once the scratch exists)
                m.src_of_job = const_cast<int32_t*>(job_src[j]);
                m.range_min = grid[j].range_min; m.resolution = grid[j].resolution;
                m.n_lines = (int32_t)lines[l]->n;
                m.job_first = jf; m.job_count = l1(j) - l0(j);
                tmp_lines += (size_t)lines[l]->n;
                ml_max = std::max(ml_max, m.n_lines);
                ml.push_back(m);
            }
        }
        const size_t list_bytes = (ml.size() * sizeof(MergeList) + 255) & ~(size_t)255;
        if ((rc = arena_reserve(ctx, ctx->merge_tmp, list_bytes + std::max<size_t>(tmp_lines, 1) * sizeof(int32_t)))) return rc;
        int32_t* tmp_base = (int32_t*)((char*)ctx->merge_tmp.ptr + list_bytes);
        for (size_t i = 0; i < ml.size(); ++i) ml[i].tmp_cidx = tmp_base + tmp_off[i];
        void* pinned = nullptr;
        if ((rc = stage_alloc(ctx, list_bytes, &pinned))) return rc;
        memcpy(pinned, ml.data(), ml.size() * sizeof(MergeList));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->merge_tmp.ptr, pinned, ml.size() * sizeof(MergeList), hipMemcpyHostToDevice, ctx->stream));
        launch_merge_ranks((const MergeList*)ctx->merge_tmp.ptr, (int)ml.size(), ml_max, ctx->stream);
        HIP_TRY(ctx, hipGetLastError());
        sc->merge_pending = false;
    }
    hipEvent_t ev = prof_begin(ctx, PROF_PREP);
    if (n_merged_jobs < n_jobs) launch_line_prep(dp, n_lists, max_lines, ctx->stream);      // (lists of merged jobs return at once)
    if (n_merged_jobs > 0)
        launch_line_prep_merged(dp, (const MergedPrep*)(d_desc + (size_t)n_lists * sizeof(PrepJob)), n_merged_jobs, (int)max_job_lines, ctx->stream);
    prof_end(ctx, PROF_PREP, ev);
    HIP_TRY(ctx, hipGetLastError());
    ctx->last_jobs = n_lists;
    ctx->last_prep_desc = dp;
    if (prep_only) return LBL_OK;
    // schedules that have not been built yet: span tables from the centre indices K1 has just written, tile costs and
    // the dispatch order, all in stream ahead of the accumulate launches that read them (nothing comes back to the host)
    for (Group& g : groups) {
        lbl_ctx::Schedule* sc = g.sched;
        if (!sc || !sc->pending) continue;
        TraceScope tr("schedule build (enqueue)", (long long)sc->total);
        const size_t jobs_bytes = ((size_t)g.count * sizeof(SchedJob) + 255) & ~(size_t)255;
        if ((rc = arena_reserve(ctx, ctx->sched, jobs_bytes + sched_scratch_bytes(sc->total)))) return rc;
        void* pinned = nullptr;
        if ((rc = stage_alloc(ctx, jobs_bytes, &pinned))) return rc;
        SchedJob* hj = (SchedJob*)pinned;
        for (int k = g.first; k < g.first + g.count; ++k) {
            const AccumJob& a = ha[k];
            SchedJob& sj = hj[k - g.first];
            sj.cidx = a.cidx; sj.n_lines = a.n_lines; sj.H = a.H; sj.p_begin = a.p_begin; sj.p_end = a.p_end;
            sj.span_first = sc->span_first[(size_t)(k - g.first)]; sj.tile_first = sc->tile_first[(size_t)(k - g.first)];
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->sched.ptr, pinned, (size_t)g.count * sizeof(SchedJob), hipMemcpyHostToDevice, ctx->stream));
        launch_schedule_build((const SchedJob*)ctx->sched.ptr, g.count, sc->total_spans, sc->total, sc->R, sc->spans_per_tile,
                              sc->far_reach, sc->cost_near, sc->cost_edge, sc->cost_far, sc->cost_fixed,
                              ctx->n_cu > 0 ? ctx->n_cu : 256, sc->d_tabs, (char*)ctx->sched.ptr + jobs_bytes, sc->d_list, ctx->stream,
                              sc->xcd_chunks, sc->xcd_pack != 0, sc->sr_chunks, sc->xcd_tol, sc->xcd_pack == 2 ? 1 : sc->xcd_pack == 3 ? 0 : -1);
        HIP_TRY(ctx, hipGetLastError());
        sc->pending = false;
    }
    // software pipeline of two contexts (lbl_ctx_chain_accumulate): this step's accumulate kernels start after the
    // predecessor's; its line prep (above) did not wait
    if (ctx->chain_pred && !ctx->capturing) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->chain_pred->accum_done, 0));
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        const Group& g = groups[gi];
        ev = prof_begin(ctx, PROF_ACCUM);
        if (balanced) {
            const size_t S = (size_t)group_spans[gi];
            char* base = (char*)ctx->bal.ptr + bal_off[gi];
            SpanRec* spans = (SpanRec*)base;
            unsigned long long* prefix = (unsigned long long*)(base + S * sizeof(SpanRec));       // 8-byte aligned: 16 B * S
            unsigned int* cnts = (unsigned int*)(base + S * sizeof(SpanRec) + (S + 1) * sizeof(unsigned long long));
            double* slab = (double*)(base + ((S * sizeof(SpanRec) + S * sizeof(unsigned int) + (S + 1) * sizeof(unsigned long long) + 255) & ~(size_t)255));
            launch_accumulate_balanced(da + g.first, g.count, (int)S, g.R, group_workers[gi], spans, cnts, prefix, slab,
                                       ctx->stream);
        } else if (g.variant == 6) {
            launch_accumulate_skew(da + g.first, g.count, g.max_tiles, g.R, g.worklist, g.total_tiles, ctx->stream, g.LS);
        } else {
            launch_accumulate(da + g.first, g.count, g.max_tiles, g.R, g.LS, g.variant, g.worklist, g.total_tiles,
                              ctx->stream, ctx->accuracy, g.grun);
        }
        prof_end(ctx, PROF_ACCUM, ev);
        HIP_TRY(ctx, hipGetLastError());
    }
    if (ctx->accum_done && !ctx->capturing) HIP_TRY(ctx, hipEventRecord(ctx->accum_done, ctx->stream));
    for (int j = 0; j < n_jobs; ++j) {
        if (!needs_regrid(grid[j])) continue;
        ev = prof_begin(ctx, PROF_REGRID);
        launch_regrid((const double*)ctx->work.ptr + work_off[j], grid[j].n_work, out_dev[j], grid[j].n_base,
                      grid[j].range_min, grid[j].range_max, ctx->stream);
        prof_end(ctx, PROF_REGRID, ev);
        HIP_TRY(ctx, hipGetLastError());
    }
    return LBL_OK;
}

// Introspection for tests: the k-th most recently used schedule of the context (0 = the last one handed out):
// its (job, tile) dispatch list and its span table, copied to the host after the stream has drained.
extern "C" int lbl_schedule_export(lbl_ctx* ctx, int k, int32_t* list, int64_t list_cap, int32_t* tabs, int64_t tabs_cap,
                                   int64_t* n_items, int64_t* n_tab_ints, int32_t* built_on_device) try {
    if (!ctx || !n_items || !n_tab_ints) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (k < 0 || (size_t)k >= ctx->schedules.size()) return fail(ctx, LBL_ERR_BAD_ARG, "no such schedule (the cache holds %d)", (int)ctx->schedules.size());
    const lbl_ctx::Schedule& S = *ctx->schedules[ctx->schedules.size() - 1 - (size_t)k];
    if (S.pending) return fail(ctx, LBL_ERR_STATE, "the schedule has not been built yet (no accumulate batch has used it)");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_tabs = (int64_t)std::abs(S.total_spans) * 8;
    *n_items = S.launch_total;
    *n_tab_ints = n_tabs;
    if (built_on_device) *built_on_device = S.total_spans > 0 ? 1 : 0;
    if (list) {
        if (list_cap < 2 * (int64_t)S.launch_total) return fail(ctx, LBL_ERR_BAD_ARG, "list too short");
        if (S.launch_total > 0) HIP_TRY(ctx, hipMemcpy(list, S.d_list, (size_t)S.launch_total * sizeof(int2), hipMemcpyDeviceToHost));
    }
    if (tabs && n_tabs > 0) {
        if (tabs_cap < n_tabs) return fail(ctx, LBL_ERR_BAD_ARG, "tabs too short");
        HIP_TRY(ctx, hipMemcpy(tabs, S.d_tabs, (size_t)n_tabs * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_xsec_accumulate_dev(lbl_ctx* ctx, int n_jobs, lbl_lines* const* lines, const lbl_iso_params* iso,
                                       const lbl_grid* grid, lbl_buffer* const* out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_jobs < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative job count");
    if (n_jobs > LBL_MAX_JOBS) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d jobs per batch", LBL_MAX_JOBS);
    if (n_jobs == 0) return LBL_OK;
    if (!lines || !iso || !grid || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    std::vector<double*> outs(n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        if (!out[j] || out[j]->ctx != ctx) return fail(ctx, LBL_ERR_STATE, "job %d: output buffer missing or from another context", j);
        if (out[j]->n < grid[j].n_base) return fail(ctx, LBL_ERR_BAD_ARG, "job %d: output buffer shorter than n_base", j);
        outs[j] = out[j]->d;
    }
    return enqueue_accumulate(ctx, n_jobs, lines, iso, grid, outs.data(), false);
} LBL_GUARD_END(ctx)

extern "C" int lbl_last_regime_counts(lbl_ctx* ctx, int n_jobs, int64_t* counts) try {
    if (!ctx || !counts) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (n_jobs < 0 || n_jobs > ctx->last_jobs) return fail(ctx, LBL_ERR_BAD_ARG, "n_jobs exceeds the last batch (%d)", ctx->last_jobs);
    if (n_jobs == 0) return LBL_OK;
    const int bpj = ctx->last_blocks_per_job;
    std::vector<unsigned int> host((size_t)n_jobs * std::max(bpj, 1) * 3, 0u);
    if (bpj > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(host.data(), ctx->counts.ptr, host.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < n_jobs; ++j) {
        int64_t c[3] = {0, 0, 0};
        const int nb = ctx->last_list_blocks[j];                    // blocks past the list's (or its merged job's) lines never ran
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 3; ++k) c[k] += host[((size_t)j * bpj + b) * 3 + k];
        counts[3 * j] = c[0]; counts[3 * j + 1] = c[1]; counts[3 * j + 2] = c[2];
    }
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_xsec_accumulate(lbl_ctx* ctx, const double* nu, const double* sw, const double* elower,
                                   const double* gamma_air, const double* gamma_self, const double* n_air,
                                   const double* delta_air, int64_t n_lines, const lbl_iso_params* iso,
                                   const lbl_grid* grid, double* xsec_out, int64_t regime_counts[3]) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (!iso || !grid || !xsec_out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    int rc = check_grid(ctx, grid);
    if (rc) return rc;
    lbl_lines* L = nullptr;
    lbl_buffer* B = nullptr;
    rc = lbl_lines_create(ctx, nu, sw, elower, gamma_air, gamma_self, n_air, delta_air, n_lines, &L);
    if (rc) return rc;
    rc = lbl_buffer_create(ctx, grid->n_base, &B);
    const int keep_lpt = ctx->lpt;
    ctx->lpt = 0;            // a one-shot line list is never seen again: no point in building (and caching) a schedule
    if (!rc) rc = lbl_xsec_accumulate_dev(ctx, 1, &L, iso, grid, &B);
    ctx->lpt = keep_lpt;
    if (!rc) rc = lbl_buffer_download(B, xsec_out, grid->n_base, 0);
    if (!rc && regime_counts) rc = lbl_last_regime_counts(ctx, 1, regime_counts);
    std::string keep = ctx->err;
    lbl_buffer_destroy(B);
    lbl_lines_destroy(L);
    if (rc) ctx->err = keep;
    return rc;
} LBL_GUARD_END(ctx)

extern "C" int lbl_line_quantities(lbl_ctx* ctx, lbl_lines* lines, const lbl_iso_params* iso, const lbl_grid* grid,
                                   int64_t* index, double* lorentz_hw, double* gauss_hw, double* intensity,
                                   int32_t* regime) try {
    if (!ctx || !lines || !iso || !grid) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    const size_t n = (size_t)lines->n;
    if (n == 0) return LBL_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    char* d = nullptr;
    const size_t bytes = n * (8 + 8 + 8 + 8 + 4);
    HIP_TRY(ctx, hipMalloc((void**)&d, bytes));
    long long* d_index = (long long*)d;
    double* d_lhw = (double*)(d + 8 * n);
    double* d_ghw = (double*)(d + 16 * n);
    double* d_inten = (double*)(d + 24 * n);
    int32_t* d_regime = (int32_t*)(d + 32 * n);
    double* no_out = nullptr;
    // the real line prep (K1) of this one job, then a reporting kernel: the centre indices are the ones K1 wrote
    int rc = enqueue_accumulate(ctx, 1, &lines, iso, grid, &no_out, true);
    hipError_t e = hipSuccess;
    if (!rc) {
        launch_line_quantities(ctx->last_prep_desc, (int)n, d_index, d_lhw, d_ghw, d_inten, d_regime, ctx->stream);
        e = hipGetLastError();
        if (e == hipSuccess && index) e = hipMemcpyAsync(index, d_index, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && lorentz_hw) e = hipMemcpyAsync(lorentz_hw, d_lhw, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && gauss_hw) e = hipMemcpyAsync(gauss_hw, d_ghw, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && intensity) e = hipMemcpyAsync(intensity, d_inten, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && regime) e = hipMemcpyAsync(regime, d_regime, 4 * n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, LBL_ERR_HIP, "line_quantities: %s", hipGetErrorString(e));
    return LBL_OK;
} LBL_GUARD_END(ctx)

// ----------------------------------------------------------------------------------------
// sweeps
// ----------------------------------------------------------------------------------------
static void planck_constants(double* pa, double* pb) {
    *pa = 2E8 * hPlanck * (cLight * cLight);      // 2E8 * h * c**2   (pyradPlanck.py:41)
    *pb = 100 * hPlanck * cLight;                 // 100 * h * c      (pyradPlanck.py:42)
}

// RN(1/c) for the kernels' div_uniform, or 0 where its exactness proof does not hold (significand of c
// all ones, c or 1/c not a normal number): the kernels then take the general divide
static double uniform_rcp(double c) {
    if (!std::isfinite(c) || c == 0.0) return 0.0;
    uint64_t bits;
    memcpy(&bits, &c, 8);
    if ((bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) return 0.0;
    const double rc = 1.0 / c;
    if (!std::isnormal(c) || !std::isnormal(rc)) return 0.0;
    return rc;
}

// budget mode (lbl_set_option "accuracy" 1): the per-molecule factor of pyradClasses.py:583 and the Planck exponent's
// per-layer factor of pyradPlanck.py:42, evaluated once on the host in the reference's left-to-right order
static double budget_factor(double conc, double P, double T) { return conc * P / 1E4 / kB / T; }
// (T = 0: +inf, so that B = a / (exp(inf) - 1) = 0 as in the reference; the entry points refuse T <= 0 for layers, a surface
// temperature of 0 means "not given")
static double budget_pbkT(double T) { const double pb = 100 * hPlanck * cLight; return pb / kB / T; }

static double axis_step(double lo, double hi, int64_t n) { return n > 1 ? (hi - lo) / (double)(n - 1) : 0.0; }

static int check_buf(lbl_ctx* ctx, const lbl_buffer* b, int64_t n, const char* what, bool required) {
    if (!b) return required ? fail(ctx, LBL_ERR_BAD_ARG, "%s is NULL", what) : LBL_OK;
    if (b->ctx != ctx) return fail(ctx, LBL_ERR_STATE, "%s belongs to another context", what);
    if (b->n < n) return fail(ctx, LBL_ERR_BAD_ARG, "%s shorter than n", what);
    return LBL_OK;
}

// smallest / largest Planck-exponent factor 100 h c / k / T of a column's terms (ColumnStepArgs.pbkT_min / _max)
static void column_pbkT_range(ColumnStepArgs* a) {
    a->pbkT_min = a->pbkT_max = a->n_terms > 0 ? a->term_pbkT[0] : 0.0;
    for (int t = 1; t < a->n_terms; ++t) {
        a->pbkT_min = std::min(a->pbkT_min, a->term_pbkT[t]);
        a->pbkT_max = std::max(a->pbkT_max, a->term_pbkT[t]);
    }
}

// A layer with more cross-section arrays than SweepArgs holds (kMaxIso): the same sums, in the same order, by the column-step
// kernel on a column of one layer (its term list lives in device memory: up to kMaxColumnIso - 1 arrays).  The reference sums
// however many molecules and isotopologues a layer holds (pyradClasses.py:566-571, 707-712).
static int layer_sweep_as_column(lbl_ctx* ctx, int n_iso, lbl_buffer* const* xsec, const int32_t* iso_mol, int n_mol,
                                 const double* conc, double P, double T, double depth, double range_min, double range_max,
                                 int64_t n, int64_t first, int64_t count, lbl_buffer* I_in, double surface_T,
                                 lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out) {
    int rc;
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if ((rc = check_buf(ctx, abs_coef, n, "abs_coef", false))) return rc;
    if ((rc = check_buf(ctx, trans, n, "trans", false))) return rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", false))) return rc;
    if (I_out && !I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "I_out needs I_in or surface_T > 0");
    std::vector<char> blk(sizeof(ColumnStepArgs), 0);
    ColumnStepArgs* a = (ColumnStepArgs*)blk.data();
    for (int i = 0; i < n_iso; ++i) {
        if ((rc = check_buf(ctx, xsec[i], n, "xsec", true))) return rc;
        if (iso_mol[i] < 0 || iso_mol[i] >= n_mol || (i > 0 && iso_mol[i] < iso_mol[i - 1]))
            return fail(ctx, LBL_ERR_BAD_ARG, "iso_mol must be non-decreasing and < n_mol");
        a->xsec[i] = xsec[i]->d;
        a->term_conc[i] = conc[iso_mol[i]];
        a->term_factor[i] = budget_factor(conc[iso_mol[i]], P, T);
        a->term_flags[i] = (i == n_iso - 1 || iso_mol[i + 1] != iso_mol[i]) ? TERM_LAST_MOL : 0;
        a->term_P[i] = P; a->term_T[i] = T; a->term_depth[i] = depth;
        a->term_rT[i] = uniform_rcp(T);
        a->term_pbkT[i] = budget_pbkT(T);
    }
    a->term_flags[n_iso - 1] |= TERM_LAST_LAYER;
    a->n_terms = n_iso; a->n_layers = 1;
    column_pbkT_range(a);
    a->ablate = ctx->ablate;
    if (abs_coef) { a->abs_coef[0] = abs_coef->d; a->layer_arrays = 1; }
    if (trans) { a->trans[0] = trans->d; a->layer_arrays = 1; }
    a->start = range_min; a->stop = range_max; a->step = axis_step(range_min, range_max, n);
    planck_constants(&a->pa, &a->pb);
    const double ts = (I_in || surface_T > 0) ? surface_T : T;       // (no radiance wanted: any positive temperature will do)
    a->surface_T = ts;
    a->r_surface_T = uniform_rcp(ts);
    a->pbk_surface = budget_pbkT(ts);
    a->I_in = I_in ? I_in->d : nullptr;
    if (I_out) {
        a->I_out = I_out->d;
    } else {                                                          // the kernel always writes the radiance: into scratch
        if ((rc = arena_reserve(ctx, ctx->ktmp, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        a->I_out = (double*)ctx->ktmp.ptr;
    }
    a->n = n; a->first = first; a->count = count;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_args = nullptr;
    if ((rc = device_args(ctx, a, sizeof(ColumnStepArgs), &d_args))) return rc;
    hipEvent_t ev = prof_begin(ctx, PROF_SWEEP);
    launch_column_step((const ColumnStepArgs*)d_args, first, count, ctx->stream, !ctx->sweep_ieee);
    prof_end(ctx, PROF_SWEEP, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
}

extern "C" int lbl_layer_sweep_dev(lbl_ctx* ctx, int n_iso, lbl_buffer* const* xsec, const int32_t* iso_mol, int n_mol,
                                   const double* conc, double P, double T, double depth, double range_min,
                                   double range_max, int64_t n, int64_t first, int64_t count, lbl_buffer* I_in,
                                   double surface_T, lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_iso < 0 || n_iso >= kMaxColumnIso || n_mol < 0 || n_mol >= kMaxColumnIso) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d isotopologues per sweep", kMaxColumnIso - 1);
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (first < 0 || count < 0 || count > n - first) return fail(ctx, LBL_ERR_BAD_ARG, "swept range outside [0, n)");
    if (count == 0) { first = 0; count = n; }
    if (n_iso > 0 && (!xsec || !iso_mol)) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (n_mol > 0 && !conc) return fail(ctx, LBL_ERR_BAD_ARG, "conc is NULL");
    if (!(T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "T must be > 0");
    if (n_iso > kMaxIso)       // more arrays than a sweep's argument block holds: the column-step kernel on a column of this one layer
        return layer_sweep_as_column(ctx, n_iso, xsec, iso_mol, n_mol, conc, P, T, depth, range_min, range_max, n, first, count,
                                     I_in, surface_T, abs_coef, trans, I_out);
    SweepArgs a;
    memset(&a, 0, sizeof a);
    int rc;
    for (int i = 0; i < n_iso; ++i) {
        if ((rc = check_buf(ctx, xsec[i], n, "xsec", true))) return rc;
        if (iso_mol[i] < 0 || iso_mol[i] >= n_mol || (i > 0 && iso_mol[i] < iso_mol[i - 1]))
            return fail(ctx, LBL_ERR_BAD_ARG, "iso_mol must be non-decreasing and < n_mol");
        a.xsec[i] = xsec[i]->d;
        a.term_conc[i] = conc[iso_mol[i]];
        a.term_factor[i] = budget_factor(conc[iso_mol[i]], P, T);
        a.term_flags[i] = (i == n_iso - 1 || iso_mol[i + 1] != iso_mol[i]) ? TERM_LAST_MOL : 0;
    }
    a.budget = !ctx->sweep_ieee; a.pbkT = budget_pbkT(T); a.pbk_surface = budget_pbkT(surface_T);
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if ((rc = check_buf(ctx, abs_coef, n, "abs_coef", false))) return rc;
    if ((rc = check_buf(ctx, trans, n, "trans", false))) return rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", false))) return rc;
    if (I_out && !I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "I_out needs I_in or surface_T > 0");
    a.n_iso = n_iso; a.n_mol = n_mol; a.P = P; a.T = T; a.depth = depth;
    a.variant = (ctx->ablate & 64) ? 0 : 1;          // streaming loads and stores: every array is touched once (-4 %)
    a.rT = uniform_rcp(T); a.r_surface_T = uniform_rcp(surface_T);
    a.start = range_min; a.stop = range_max; a.step = axis_step(range_min, range_max, n);
    planck_constants(&a.pa, &a.pb);
    a.surface_T = surface_T;
    a.I_in = I_in ? I_in->d : nullptr;
    a.abs_coef = abs_coef ? abs_coef->d : nullptr;
    a.trans = trans ? trans->d : nullptr;
    a.I_out = I_out ? I_out->d : nullptr;
    a.n = n; a.first = first; a.count = count;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev = prof_begin(ctx, PROF_SWEEP);
    launch_layer_sweep(a, ctx->stream);
    prof_end(ctx, PROF_SWEEP, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_layer_step_dev(lbl_ctx* ctx, int n_iso, lbl_lines* const* lines, const lbl_iso_params* iso,
                                  const lbl_grid* grid, lbl_buffer* const* xsec, const int32_t* iso_mol, int n_mol,
                                  const double* conc, double depth, lbl_buffer* I_in, double surface_T,
                                  lbl_buffer* abs_coef, lbl_buffer* trans, lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_iso < 1 || n_iso >= kMaxColumnIso || n_mol < 1 || n_mol >= kMaxColumnIso) return fail(ctx, LBL_ERR_BAD_ARG, "1..%d line lists and molecules per layer step", kMaxColumnIso - 1);
    if (!lines || !iso || !grid || !xsec || !iso_mol || !conc) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    int rc;
    if ((rc = check_grid(ctx, grid))) return rc;
    const int64_t n = grid->n_base;
    std::vector<double*> outs((size_t)n_iso);
    std::vector<lbl_grid> grids((size_t)n_iso, *grid);
    for (int i = 0; i < n_iso; ++i) {
        if ((rc = check_buf(ctx, xsec[i], n, "xsec", true))) return rc;
        if (iso_mol[i] < 0 || iso_mol[i] >= n_mol || (i > 0 && iso_mol[i] < iso_mol[i - 1]))
            return fail(ctx, LBL_ERR_BAD_ARG, "iso_mol must be non-decreasing and < n_mol");
        if (iso[i].T != iso[0].T || iso[i].P != iso[0].P)
            return fail(ctx, LBL_ERR_BAD_ARG, "line list %d: T and P must be the layer's (those of line list 0)", i);
        outs[i] = xsec[i]->d;
    }
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if ((rc = check_buf(ctx, abs_coef, n, "abs_coef", false))) return rc;
    if ((rc = check_buf(ctx, trans, n, "trans", false))) return rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", false))) return rc;
    if (I_out && !I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "I_out needs I_in or surface_T > 0");
    // One line list on the base grid: the sweep of a point runs in the accumulate kernel's output stage.
    // (A molecule without line lists before it adds 0 * conc to k in the sweep kernel: same bits.)
    const bool fusable = n_iso == 1 && (ctx->accum_variant == 3 || ctx->accum_variant == 5) && !needs_regrid(*grid) &&
                         !ctx->no_fuse;
    if (fusable) {
        FusedSweep f;
        memset(&f, 0, sizeof f);
        f.P = iso[0].P; f.T = iso[0].T; f.depth = depth;
        f.rT = uniform_rcp(f.T); f.r_surface_T = uniform_rcp(surface_T);
        f.start = grid->range_min; f.stop = grid->range_max; f.step = axis_step(grid->range_min, grid->range_max, n);
        planck_constants(&f.pa, &f.pb);
        f.surface_T = surface_T;
        f.I_in = I_in ? I_in->d : nullptr;
        f.abs_coef = abs_coef ? abs_coef->d : nullptr;
        f.trans = trans ? trans->d : nullptr;
        f.I_out = I_out ? I_out->d : nullptr;
        f.n = n; f.on = 1;
        f.budget = !ctx->sweep_ieee; f.factor = budget_factor(conc[iso_mol[0]], f.P, f.T);
        f.pbkT = budget_pbkT(f.T); f.pbk_surface = budget_pbkT(surface_T);
        return enqueue_accumulate(ctx, 1, lines, iso, grids.data(), outs.data(), false, &f, conc[iso_mol[0]]);
    }
    // several line lists, a work grid that needs the regrid kernel, or a kernel variant without the fused
    // stage: the accumulate launch and the sweep launch
    if ((rc = enqueue_accumulate(ctx, n_iso, lines, iso, grids.data(), outs.data(), false))) return rc;
    long long sf, sc;
    shard_range(*grid, &sf, &sc);
    const bool whole = sc == grid->n_work;
    return lbl_layer_sweep_dev(ctx, n_iso, xsec, iso_mol, n_mol, conc, iso[0].P, iso[0].T, depth, grid->range_min,
                               grid->range_max, n, whole ? 0 : sf, whole ? 0 : sc, I_in, surface_T, abs_coef, trans, I_out);
} LBL_GUARD_END(ctx)

// ---- merged layer jobs: north_star's "shared wavenumber-grid absorption-coefficient array" -------------------------------
// The reference's layer result is Layer.absCoef = sum over molecules of Molecule.absCoef = (sum over isotopologues of
// crossSection) * concentration * P / 1E4 / k / T (pyradClasses.py:707-712, 581-583, 566-571).  Here every line of every
// list of the layer accumulates straight into that sum: K1 multiplies each list's amplitudes by its molecule's factor
// f_m = conc_m P / 1E4 / k / T (evaluated on the host in the reference's order) and writes all records into ONE array in
// centre-index order; K2 runs ONE job over it.  The per-isotopologue cross sections are not materialised (the reference
// computes them lazily too, progressCrossSection, pyradClasses.py:32-88: lbl_xsec_accumulate_dev produces any of them on
// demand).  So that K2's intermediate magnitudes stay those of a cross section (the running fraction's range analysis), the
// weights are f_m / 2^e with 2^e the power of two just above the largest |f_m|, and every sum is multiplied by 2^e on its way
// out: both scalings are exact.
static void merged_weights(int n_iso, const int32_t* iso_mol, const double* conc, double P, double T, double* weight,
                           double* out_scale) {
    double fmax = 0.0;
    bool finite = true;
    for (int i = 0; i < n_iso; ++i) {
        weight[i] = budget_factor(conc[iso_mol[i]], P, T);
        finite = finite && std::isfinite(weight[i]);
        fmax = std::max(fmax, std::fabs(weight[i]));
    }
    int e = 0;
    if (finite && fmax > 0.0) (void)std::frexp(fmax, &e);       // fmax = m 2^e, m in [0.5, 1)
    if (e > 1000 || e < -1000) e = 0;                            // (absurd inputs: leave the magnitudes alone)
    *out_scale = std::ldexp(1.0, e);
    for (int i = 0; i < n_iso; ++i) weight[i] = std::ldexp(weight[i], -e);
}

static int check_layer_lists(lbl_ctx* ctx, int n_iso, const lbl_iso_params* iso, const int32_t* iso_mol, int n_mol, int layer) {
    for (int i = 0; i < n_iso; ++i) {
        if (iso_mol[i] < 0 || iso_mol[i] >= n_mol || (i > 0 && iso_mol[i] < iso_mol[i - 1]))
            return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: iso_mol must be non-decreasing and < n_mol", layer);
        if (iso[i].T != iso[0].T || iso[i].P != iso[0].P)
            return fail(ctx, LBL_ERR_BAD_ARG, "layer %d, line list %d: T and P must be the layer's (those of its first line list)", layer, i);
    }
    return LBL_OK;
}

extern "C" int lbl_layer_merged_step_dev(lbl_ctx* ctx, int n_iso, lbl_lines* const* lines, const lbl_iso_params* iso,
                                         const lbl_grid* grid, const int32_t* iso_mol, int n_mol, const double* conc,
                                         double depth, lbl_buffer* I_in, double surface_T, lbl_buffer* abs_coef,
                                         lbl_buffer* trans, lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_iso < 1 || n_iso > kMaxIso || n_mol < 1 || n_mol > kMaxIso) return fail(ctx, LBL_ERR_BAD_ARG, "1..%d line lists and molecules per layer step", kMaxIso);
    if (!lines || !iso || !grid || !iso_mol || !conc) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (ctx->sweep_ieee) return fail(ctx, LBL_ERR_BAD_ARG, "merged layer jobs exist in the sweeps' default arithmetic only: with \"sweep_ieee_divisions\" 1 use the per-line-list step");
    int rc;
    if ((rc = check_grid(ctx, grid))) return rc;
    if ((rc = check_layer_lists(ctx, n_iso, iso, iso_mol, n_mol, 0))) return rc;
    const int64_t n = grid->n_base;
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if ((rc = check_buf(ctx, abs_coef, n, "abs_coef", false))) return rc;
    if ((rc = check_buf(ctx, trans, n, "trans", false))) return rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", false))) return rc;
    if (I_out && !I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "I_out needs I_in or surface_T > 0");
    if (!abs_coef && !trans && !I_out) return fail(ctx, LBL_ERR_BAD_ARG, "no output array");
    std::vector<double> weight((size_t)n_iso);
    double out_scale = 1.0;
    merged_weights(n_iso, iso_mol, conc, iso[0].P, iso[0].T, weight.data(), &out_scale);
    const int first[2] = {0, n_iso};
    MergeSpec ms{first, weight.data(), &out_scale};
    FusedSweep f;
    memset(&f, 0, sizeof f);
    f.P = iso[0].P; f.T = iso[0].T; f.depth = depth;
    f.rT = uniform_rcp(f.T); f.r_surface_T = uniform_rcp(surface_T);
    f.start = grid->range_min; f.stop = grid->range_max; f.step = axis_step(grid->range_min, grid->range_max, n);
    planck_constants(&f.pa, &f.pb);
    f.surface_T = surface_T;
    f.I_in = I_in ? I_in->d : nullptr;
    f.abs_coef = abs_coef ? abs_coef->d : nullptr;
    f.trans = trans ? trans->d : nullptr;
    f.I_out = I_out ? I_out->d : nullptr;
    f.n = n; f.on = 2;
    f.budget = 1; f.factor = 1.0;                  // (merged jobs exist in the sweeps' default arithmetic only)
    f.pbkT = budget_pbkT(f.T); f.pbk_surface = budget_pbkT(surface_T);
    if (!needs_regrid(*grid)) {
        double* out = nullptr;
        if (!trans && !I_out) {
            // only the absorption coefficient is wanted: the kernel's plain store, no sweep arithmetic
            out = abs_coef->d;
            return enqueue_accumulate(ctx, 1, lines, iso, grid, &out, false, nullptr, 0.0, &ms);
        }
        return enqueue_accumulate(ctx, 1, lines, iso, grid, &out, false, &f, 0.0, &ms);
    }
    // a coarser work grid (dynamic resolution, pyradClasses.py:659-662, 401-405): k on the work grid, np.interp onto the base
    // grid (linear, like the sum it is applied to), then the sweep of that one array
    double* kdst = abs_coef ? abs_coef->d : nullptr;
    if (!kdst) {
        if ((rc = arena_reserve(ctx, ctx->ktmp, (size_t)std::max<int64_t>(n, 1) * sizeof(double)))) return rc;
        kdst = (double*)ctx->ktmp.ptr;
    }
    if ((rc = enqueue_accumulate(ctx, 1, lines, iso, grid, &kdst, false, nullptr, 0.0, &ms))) return rc;
    if (!trans && !I_out) return LBL_OK;
    SweepArgs a;
    memset(&a, 0, sizeof a);
    a.xsec[0] = kdst; a.term_conc[0] = 1.0; a.term_factor[0] = 1.0; a.term_flags[0] = TERM_LAST_MOL;
    a.budget = 1; a.pbkT = f.pbkT; a.pbk_surface = f.pbk_surface;
    a.n_iso = 1; a.n_mol = 1; a.P = f.P; a.T = f.T; a.depth = depth;
    a.variant = 0;                                   // (k is read here and again by the caller: no streaming hints)
    a.rT = f.rT; a.r_surface_T = f.r_surface_T;
    a.start = f.start; a.stop = f.stop; a.step = f.step; a.pa = f.pa; a.pb = f.pb;
    a.surface_T = surface_T;
    a.I_in = f.I_in; a.abs_coef = nullptr; a.trans = f.trans; a.I_out = f.I_out;
    a.n = n; a.first = 0; a.count = n;
    hipEvent_t ev = prof_begin(ctx, PROF_SWEEP);
    launch_layer_sweep(a, ctx->stream);
    prof_end(ctx, PROF_SWEEP, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_layers_merged_accumulate_dev(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_lines* const* lines,
                                                const lbl_iso_params* iso, const lbl_grid* grid, const int32_t* iso_mol,
                                                const int32_t* n_mol, const double* conc, lbl_buffer* const* abs_coef) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_layers < 0 || n_layers > LBL_MAX_JOBS) return fail(ctx, LBL_ERR_BAD_ARG, "0..%d layers per batch", LBL_MAX_JOBS);
    if (n_layers == 0) return LBL_OK;
    if (!n_iso || !lines || !iso || !grid || !iso_mol || !n_mol || !conc || !abs_coef) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    std::vector<int> first((size_t)n_layers + 1, 0);
    std::vector<double> out_scale((size_t)n_layers, 1.0);
    std::vector<double*> outs((size_t)n_layers);
    long long total = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (n_iso[l] < 1 || n_iso[l] > kMaxIso || n_mol[l] < 1 || n_mol[l] > kMaxIso)
            return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: 1..%d line lists and molecules per layer", l, kMaxIso);
        total += n_iso[l];
        if (total > LBL_MAX_JOBS) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d line lists per batch", LBL_MAX_JOBS);
        first[(size_t)l + 1] = (int)total;
    }
    std::vector<double> weight((size_t)total);
    int rc, mol0 = 0;
    for (int l = 0; l < n_layers; ++l) {
        const int i0 = first[(size_t)l];
        if ((rc = check_grid(ctx, &grid[l]))) return rc;
        if ((rc = check_layer_lists(ctx, n_iso[l], iso + i0, iso_mol + i0, n_mol[l], l))) return rc;
        if ((rc = check_buf(ctx, abs_coef[l], grid[l].n_base, "abs_coef", true))) return rc;
        merged_weights(n_iso[l], iso_mol + i0, conc + mol0, iso[i0].P, iso[i0].T, weight.data() + i0, &out_scale[(size_t)l]);
        outs[(size_t)l] = abs_coef[l]->d;
        mol0 += n_mol[l];
    }
    MergeSpec ms{first.data(), weight.data(), out_scale.data()};
    return enqueue_accumulate(ctx, n_layers, lines, iso, grid, outs.data(), false, nullptr, 0.0, &ms);
} LBL_GUARD_END(ctx)

extern "C" int lbl_column_fold_dev(lbl_ctx* ctx, int n_layers, lbl_buffer* const* abs_coef, const double* T,
                                   const double* depth, double range_min, double range_max, int64_t n, int64_t first,
                                   int64_t count, lbl_buffer* I_in, double surface_T, lbl_buffer* const* trans,
                                   lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_layers < 0 || n_layers > kMaxLayers) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d layers", kMaxLayers);
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (first < 0 || count < 0 || count > n - first) return fail(ctx, LBL_ERR_BAD_ARG, "swept range outside [0, n)");
    if (count == 0) { first = 0; count = n; }
    if (n_layers > 0 && (!abs_coef || !T || !depth)) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (ctx->sweep_ieee) return fail(ctx, LBL_ERR_BAD_ARG, "the fold over absorption coefficients exists in the sweeps' default arithmetic only: with \"sweep_ieee_divisions\" 1 use lbl_column_step_dev on the cross sections");
    int rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", true))) return rc;
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if (!I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "need I_in or surface_T > 0");
    std::vector<char> blk(sizeof(ColumnStepArgs), 0);
    ColumnStepArgs* a = (ColumnStepArgs*)blk.data();
    for (int l = 0; l < n_layers; ++l) {
        if ((rc = check_buf(ctx, abs_coef[l], n, "abs_coef", true))) return rc;
        if (!(T[l] > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: T must be > 0", l);
        // one term per layer: the "cross section" is the layer's absorption coefficient, its factor 1
        a->xsec[l] = abs_coef[l]->d;
        a->term_conc[l] = 1.0; a->term_factor[l] = 1.0;
        a->term_flags[l] = TERM_LAST_MOL | TERM_LAST_LAYER;
        a->term_P[l] = 0.0; a->term_T[l] = T[l]; a->term_depth[l] = depth[l];
        a->term_rT[l] = uniform_rcp(T[l]);
        a->term_pbkT[l] = budget_pbkT(T[l]);
        if (trans && trans[l]) { if ((rc = check_buf(ctx, trans[l], n, "trans", true))) return rc; a->trans[l] = trans[l]->d; a->layer_arrays = 1; }
    }
    a->n_terms = n_layers;
    column_pbkT_range(a);
    a->ablate = ctx->ablate;
    a->n_layers = n_layers;
    a->start = range_min; a->stop = range_max; a->step = axis_step(range_min, range_max, n);
    planck_constants(&a->pa, &a->pb);
    a->surface_T = surface_T;
    a->r_surface_T = uniform_rcp(surface_T);
    a->pbk_surface = budget_pbkT(surface_T);
    a->I_in = I_in ? I_in->d : nullptr;
    a->I_out = I_out->d;
    a->n = n; a->first = first; a->count = count;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_args = nullptr;
    if ((rc = device_args(ctx, a, sizeof(ColumnStepArgs), &d_args))) return rc;
    hipEvent_t ev = prof_begin(ctx, PROF_COLUMN);
    launch_column_step((const ColumnStepArgs*)d_args, first, count, ctx->stream, 1, 1);
    prof_end(ctx, PROF_COLUMN, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

// ---- resident column (ABI 5): the argument blocks of a column's merged accumulate jobs and of its fold, kept on the C side ----
// Atmosphere.transmission of a 30-layer column spent 0.45 ms of a 4.6 ms call in Python and ctypes before the first kernel was
// enqueued: 90 line-list handles, 90 parameter blocks and 30 grids marshalled per call, then four fold calls and four download
// calls.  A column handle holds those blocks between calls (lbl_column_set_layer refreshes the one layer a mutator touched); a
// call is ONE entry point: the merged accumulate jobs of the layers that are due, the fold bottom to top in `pieces` pieces of
// the grid, each piece's part of the outgoing spectrum on its way to the host (lbl_buffer_download_async) while the next piece
// is folded.  The handle owns nothing on the device: line lists and buffers stay the caller's and must outlive it.
struct lbl_column {
    lbl_ctx* ctx;
    int n_layers;
    std::vector<int32_t> n_iso, n_mol, iso_first, mol_first, iso_mol;
    std::vector<lbl_lines*> lines;
    std::vector<lbl_iso_params> iso;
    std::vector<lbl_grid> grid;
    std::vector<double> conc, depth;
    std::vector<lbl_buffer*> abs_coef;
};

extern "C" int lbl_column_create(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_lines* const* lines,
                                 const lbl_iso_params* iso, const lbl_grid* grid, const int32_t* iso_mol, const int32_t* n_mol,
                                 const double* conc, const double* depth, lbl_buffer* const* abs_coef, lbl_column** out) try {
    if (!ctx || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (n_layers < 1 || n_layers > kMaxLayers) return fail(ctx, LBL_ERR_BAD_ARG, "1..%d layers per column", kMaxLayers);
    if (!n_iso || !lines || !iso || !grid || !iso_mol || !n_mol || !conc || !depth || !abs_coef) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    std::unique_ptr<lbl_column> c(new lbl_column);
    c->ctx = ctx; c->n_layers = n_layers;
    int rc;
    long long ni = 0, nm = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (n_iso[l] < 1 || n_iso[l] > kMaxIso || n_mol[l] < 1 || n_mol[l] > kMaxIso)
            return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: 1..%d line lists and molecules per layer", l, kMaxIso);
        c->iso_first.push_back((int32_t)ni); c->mol_first.push_back((int32_t)nm);
        if ((rc = check_grid(ctx, &grid[l]))) return rc;
        if ((rc = check_layer_lists(ctx, n_iso[l], iso + ni, iso_mol + ni, n_mol[l], l))) return rc;
        if ((rc = check_buf(ctx, abs_coef[l], grid[l].n_base, "abs_coef", true))) return rc;
        if (grid[l].n_base != grid[0].n_base || grid[l].range_min != grid[0].range_min || grid[l].range_max != grid[0].range_max)
            return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: all layers of a column share one wavenumber range and base grid", l);
        for (int i = 0; i < n_iso[l]; ++i)
            if (!lines[ni + i] || lines[ni + i]->ctx != ctx) return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: line list %d is NULL or belongs to another context", l, i);
        ni += n_iso[l]; nm += n_mol[l];
    }
    c->n_iso.assign(n_iso, n_iso + n_layers); c->n_mol.assign(n_mol, n_mol + n_layers);
    c->lines.assign(lines, lines + ni); c->iso.assign(iso, iso + ni); c->iso_mol.assign(iso_mol, iso_mol + ni);
    c->grid.assign(grid, grid + n_layers); c->conc.assign(conc, conc + nm); c->depth.assign(depth, depth + n_layers);
    c->abs_coef.assign(abs_coef, abs_coef + n_layers);
    ctx->live_objects++;
    *out = c.release();
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_column_destroy(lbl_column* col) try {
    if (!col) return LBL_OK;
    col->ctx->live_objects--;
    delete col;
    return LBL_OK;
} LBL_GUARD_END(col ? col->ctx : nullptr)

extern "C" int lbl_column_set_layer(lbl_column* col, int layer, lbl_lines* const* lines, const lbl_iso_params* iso,
                                    const lbl_grid* grid, const double* conc, double depth, lbl_buffer* abs_coef) try {
    if (!col) return fail(nullptr, LBL_ERR_BAD_ARG, "column is NULL");
    lbl_ctx* ctx = col->ctx;
    if (layer < 0 || layer >= col->n_layers) return fail(ctx, LBL_ERR_BAD_ARG, "no layer %d", layer);
    if (!lines || !iso || !grid || !conc || !abs_coef) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    const int i0 = col->iso_first[(size_t)layer], m0 = col->mol_first[(size_t)layer], ni = col->n_iso[(size_t)layer];
    int rc;
    if ((rc = check_grid(ctx, grid))) return rc;
    if ((rc = check_layer_lists(ctx, ni, iso, col->iso_mol.data() + i0, col->n_mol[(size_t)layer], layer))) return rc;
    if ((rc = check_buf(ctx, abs_coef, grid->n_base, "abs_coef", true))) return rc;
    if (grid->n_base != col->grid[0].n_base && col->n_layers > 1)
        return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: all layers of a column share one base grid", layer);
    for (int i = 0; i < ni; ++i) {
        if (!lines[i] || lines[i]->ctx != ctx) return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: line list %d is NULL or belongs to another context", layer, i);
        col->lines[(size_t)(i0 + i)] = lines[i];
        col->iso[(size_t)(i0 + i)] = iso[i];
    }
    col->grid[(size_t)layer] = *grid;
    for (int m = 0; m < col->n_mol[(size_t)layer]; ++m) col->conc[(size_t)(m0 + m)] = conc[m];
    col->depth[(size_t)layer] = depth;
    col->abs_coef[(size_t)layer] = abs_coef;
    return LBL_OK;
} LBL_GUARD_END(col ? col->ctx : nullptr)

extern "C" int lbl_column_transmission(lbl_column* col, const uint8_t* due, lbl_buffer* I_in, double surface_T,
                                       lbl_buffer* I_out, double* host_out, int pieces) try {
    if (!col) return fail(nullptr, LBL_ERR_BAD_ARG, "column is NULL");
    lbl_ctx* ctx = col->ctx;
    const int nl = col->n_layers;
    const int64_t n = col->grid[0].n_base;
    if (pieces < 1 || pieces > 64) return fail(ctx, LBL_ERR_BAD_ARG, "1..64 pieces");
    int rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", true))) return rc;
    // the layers that are due: one merged accumulate job each, all in one launch sequence
    std::vector<int32_t> d_niso, d_nmol, d_isomol;
    std::vector<lbl_lines*> d_lines;
    std::vector<lbl_iso_params> d_iso;
    std::vector<lbl_grid> d_grid;
    std::vector<double> d_conc;
    std::vector<lbl_buffer*> d_k;
    for (int l = 0; l < nl; ++l) {
        if (due && !due[l]) continue;
        const size_t i0 = (size_t)col->iso_first[(size_t)l], m0 = (size_t)col->mol_first[(size_t)l];
        const size_t ni = (size_t)col->n_iso[(size_t)l], nm = (size_t)col->n_mol[(size_t)l];
        d_niso.push_back((int32_t)ni); d_nmol.push_back((int32_t)nm);
        d_lines.insert(d_lines.end(), col->lines.begin() + i0, col->lines.begin() + i0 + ni);
        d_iso.insert(d_iso.end(), col->iso.begin() + i0, col->iso.begin() + i0 + ni);
        d_isomol.insert(d_isomol.end(), col->iso_mol.begin() + i0, col->iso_mol.begin() + i0 + ni);
        d_conc.insert(d_conc.end(), col->conc.begin() + m0, col->conc.begin() + m0 + nm);
        d_grid.push_back(col->grid[(size_t)l]);
        d_k.push_back(col->abs_coef[(size_t)l]);
    }
    if (!d_niso.empty() &&
        (rc = lbl_layers_merged_accumulate_dev(ctx, (int)d_niso.size(), d_niso.data(), d_lines.data(), d_iso.data(), d_grid.data(),
                                               d_isomol.data(), d_nmol.data(), d_conc.data(), d_k.data())))
        return rc;
    std::vector<double> T((size_t)nl);
    for (int l = 0; l < nl; ++l) T[(size_t)l] = col->iso[(size_t)col->iso_first[(size_t)l]].T;
    // the fold in pieces (multiples of four points: the fold kernel's vector width), each piece's outgoing spectrum
    // downloaded beside the next piece
    const int64_t step = std::max<int64_t>((((n + pieces - 1) / pieces) + 3) & ~(int64_t)3, 4);
    for (int64_t lo = 0; lo < n; lo += step) {
        const int64_t cnt = std::min(step, n - lo);
        if ((rc = lbl_column_fold_dev(ctx, nl, col->abs_coef.data(), T.data(), col->depth.data(), col->grid[0].range_min,
                                      col->grid[0].range_max, n, lo, cnt, I_in, surface_T, nullptr, I_out)))
            return rc;
        if (host_out && (rc = lbl_buffer_download_async(I_out, host_out + lo, cnt, lo))) return rc;
    }
    return LBL_OK;
} LBL_GUARD_END(col ? col->ctx : nullptr)

extern "C" int lbl_column_sweep_dev(lbl_ctx* ctx, int n_layers, lbl_buffer* const* trans, const double* layer_T,
                                    double range_min, double range_max, int64_t n, int64_t first, int64_t count,
                                    lbl_buffer* I_in, double surface_T, lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_layers < 0 || n_layers > kMaxLayers) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d layers", kMaxLayers);
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (first < 0 || count < 0 || count > n - first) return fail(ctx, LBL_ERR_BAD_ARG, "swept range outside [0, n)");
    if (count == 0) { first = 0; count = n; }
    if (n_layers > 0 && (!trans || !layer_T)) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    int rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", true))) return rc;
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if (!I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "need I_in or surface_T > 0");
    std::vector<char> blk(sizeof(ColumnArgs), 0);
    ColumnArgs* a = (ColumnArgs*)blk.data();
    for (int l = 0; l < n_layers; ++l) {
        if ((rc = check_buf(ctx, trans[l], n, "trans", true))) return rc;
        if (!(layer_T[l] > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "layer_T must be > 0");
        a->trans[l] = trans[l]->d;
        a->layer_T[l] = layer_T[l];
        a->r_layer_T[l] = uniform_rcp(layer_T[l]);
        a->pbkT[l] = budget_pbkT(layer_T[l]);
    }
    a->n_layers = n_layers;
    a->start = range_min; a->stop = range_max; a->step = axis_step(range_min, range_max, n);
    planck_constants(&a->pa, &a->pb);
    a->surface_T = surface_T;
    a->r_surface_T = uniform_rcp(surface_T);
    a->pbk_surface = budget_pbkT(surface_T);
    a->I_in = I_in ? I_in->d : nullptr;
    a->I_out = I_out->d;
    a->n = n; a->first = first; a->count = count;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_args = nullptr;
    if ((rc = device_args(ctx, a, sizeof(ColumnArgs), &d_args))) return rc;
    hipEvent_t ev = prof_begin(ctx, PROF_COLUMN);
    launch_column_sweep((const ColumnArgs*)d_args, count, ctx->stream, !ctx->sweep_ieee);
    prof_end(ctx, PROF_COLUMN, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_column_step_dev(lbl_ctx* ctx, int n_layers, const int32_t* n_iso, lbl_buffer* const* xsec,
                                   const int32_t* iso_mol, const int32_t* n_mol, const double* conc, const double* P,
                                   const double* T, const double* depth, double range_min, double range_max, int64_t n,
                                   int64_t first, int64_t count, lbl_buffer* I_in, double surface_T,
                                   lbl_buffer* const* abs_coef, lbl_buffer* const* trans, lbl_buffer* I_out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_layers < 0 || n_layers > kMaxLayers) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d layers", kMaxLayers);
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (first < 0 || count < 0 || count > n - first) return fail(ctx, LBL_ERR_BAD_ARG, "swept range outside [0, n)");
    if (count == 0) { first = 0; count = n; }
    if (n_layers > 0 && (!n_iso || !n_mol || !P || !T || !depth)) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    int rc;
    if ((rc = check_buf(ctx, I_out, n, "I_out", true))) return rc;
    if ((rc = check_buf(ctx, I_in, n, "I_in", false))) return rc;
    if (!I_in && !(surface_T > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "need I_in or surface_T > 0");
    std::vector<char> blk(sizeof(ColumnStepArgs), 0);
    ColumnStepArgs* a = (ColumnStepArgs*)blk.data();
    int iso0 = 0, mol0 = 0, nt = 0;
    for (int l = 0; l < n_layers; ++l) {
        if (n_iso[l] < 0 || n_mol[l] < 0 || nt + n_iso[l] + 1 > kMaxColumnIso || mol0 + n_mol[l] > kMaxColumnIso)
            return fail(ctx, LBL_ERR_BAD_ARG, "at most %d isotopologues per column", kMaxColumnIso - 1);
        if (!(T[l] > 0)) return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: T must be > 0", l);
        if ((n_iso[l] > 0 && (!xsec || !iso_mol)) || (n_mol[l] > 0 && !conc)) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
        // the layer's terms: one per cross-section array, molecule after molecule (a molecule without line
        // lists adds 0 to the absorption coefficient: no term; a layer without any gets one empty term)
        for (int i = 0; i < n_iso[l]; ++i) {
            if ((rc = check_buf(ctx, xsec[iso0 + i], n, "xsec", true))) return rc;
            const int32_t m = iso_mol[iso0 + i];
            if (m < 0 || m >= n_mol[l] || (i > 0 && m < iso_mol[iso0 + i - 1]))
                return fail(ctx, LBL_ERR_BAD_ARG, "layer %d: iso_mol must be non-decreasing and < n_mol", l);
            a->xsec[nt] = xsec[iso0 + i]->d;
            a->term_conc[nt] = conc[mol0 + m];
            a->term_factor[nt] = budget_factor(conc[mol0 + m], P[l], T[l]);
            a->term_flags[nt] = (i == n_iso[l] - 1 || iso_mol[iso0 + i + 1] != m) ? TERM_LAST_MOL : 0;
            ++nt;
        }
        if (n_iso[l] == 0) {
            // a layer without line lists: one term that reads zeros (grown on first use; absorbs nothing)
            if ((rc = arena_reserve(ctx, ctx->zeros, (size_t)n * sizeof(double)))) return rc;
            if (ctx->zeros_set < ctx->zeros.cap) {
                HIP_TRY(ctx, hipMemsetAsync(ctx->zeros.ptr, 0, ctx->zeros.cap, ctx->stream));
                ctx->zeros_set = ctx->zeros.cap;
            }
            a->xsec[nt] = (const double*)ctx->zeros.ptr; a->term_conc[nt] = 0.0; a->term_flags[nt] = TERM_LAST_MOL; ++nt;
        }
        a->term_flags[nt - 1] |= TERM_LAST_LAYER;
        for (int t = nt - std::max(n_iso[l], 1); t < nt; ++t) {
            a->term_P[t] = P[l]; a->term_T[t] = T[l]; a->term_depth[t] = depth[l];
            a->term_rT[t] = uniform_rcp(T[l]);
            a->term_pbkT[t] = budget_pbkT(T[l]);
        }
        if (abs_coef && abs_coef[l]) { if ((rc = check_buf(ctx, abs_coef[l], n, "abs_coef", true))) return rc; a->abs_coef[l] = abs_coef[l]->d; a->layer_arrays = 1; }
        if (trans && trans[l]) { if ((rc = check_buf(ctx, trans[l], n, "trans", true))) return rc; a->trans[l] = trans[l]->d; a->layer_arrays = 1; }
        iso0 += n_iso[l]; mol0 += n_mol[l];
    }
    a->n_terms = nt;
    column_pbkT_range(a);
    a->ablate = ctx->ablate;
    a->n_layers = n_layers;
    a->start = range_min; a->stop = range_max; a->step = axis_step(range_min, range_max, n);
    planck_constants(&a->pa, &a->pb);
    a->surface_T = surface_T;
    a->r_surface_T = uniform_rcp(surface_T);
    a->pbk_surface = budget_pbkT(surface_T);
    a->I_in = I_in ? I_in->d : nullptr;
    a->I_out = I_out->d;
    a->n = n; a->first = first; a->count = count;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* d_args = nullptr;
    if ((rc = device_args(ctx, a, sizeof(ColumnStepArgs), &d_args))) return rc;
    hipEvent_t ev = prof_begin(ctx, PROF_COLUMN);
    launch_column_step((const ColumnStepArgs*)d_args, first, count, ctx->stream, !ctx->sweep_ieee);
    prof_end(ctx, PROF_COLUMN, ev);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_sum_dev(lbl_ctx* ctx, int n_in, lbl_buffer* const* in, int64_t n, lbl_buffer* out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n_in < 0 || n_in > kMaxIso) return fail(ctx, LBL_ERR_BAD_ARG, "at most %d inputs", kMaxIso);
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (n_in > 0 && !in) return fail(ctx, LBL_ERR_BAD_ARG, "in is NULL");
    int rc;
    if ((rc = check_buf(ctx, out, n, "out", true))) return rc;
    SumArgs a;
    memset(&a, 0, sizeof a);
    for (int i = 0; i < n_in; ++i) {
        if ((rc = check_buf(ctx, in[i], n, "in", true))) return rc;
        a.in[i] = in[i]->d;
    }
    a.n_in = n_in; a.out = out->d; a.n = n;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_sum(a, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_optical_dev(lbl_ctx* ctx, lbl_buffer* trans, int64_t n, int kind, lbl_buffer* out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if (kind < 0 || kind > 2) return fail(ctx, LBL_ERR_BAD_ARG, "kind must be 0 (emissivity), 1 (absorbance) or 2 (optical depth)");
    int rc;
    if ((rc = check_buf(ctx, trans, n, "trans", true))) return rc;
    if ((rc = check_buf(ctx, out, n, "out", true))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_optical(trans->d, n, kind, out->d, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_planck_dev(lbl_ctx* ctx, double range_min, double range_max, int64_t n, double T, lbl_buffer* out) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    int rc;
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if ((rc = check_buf(ctx, out, n, "out", true))) return rc;
    double pa, pb;
    planck_constants(&pa, &pb);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_planck(out->d, n, range_min, range_max, T, uniform_rcp(T), pa, pb, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_band_integral(lbl_ctx* ctx, lbl_buffer* spectrum, int64_t n, double unit_angle, double res,
                                 double* result) try {
    if (!ctx || !result) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    int rc;
    if (n < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative n");
    if ((rc = check_buf(ctx, spectrum, n, "spectrum", true))) return rc;
    if (n == 0) { *result = 0.0 * unit_angle * res; return LBL_OK; }
    const int nb = band_partial_count(n);
    if ((rc = arena_reserve(ctx, ctx->red, (size_t)(nb + 1) * sizeof(double)))) return rc;
    double* partial = (double*)ctx->red.ptr;
    double* dres = partial + nb;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_band_integral(spectrum->d, n, partial, dres, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    double s = 0.0;
    HIP_TRY(ctx, hipMemcpyAsync(&s, dres, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *result = s * unit_angle * res;              // value * unitAngle * res (pyradClasses.py:28)
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_line_survey_dev(lbl_ctx* ctx, lbl_lines* lines, const lbl_grid* grid, lbl_buffer* out) try {
    if (!ctx || !lines || !grid) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (lines->ctx != ctx) return fail(ctx, LBL_ERR_STATE, "line list belongs to another context");
    int rc;
    if ((rc = check_grid(ctx, grid))) return rc;
    if ((rc = check_buf(ctx, out, grid->n_base, "out", true))) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (grid->n_base > 0) HIP_TRY(ctx, hipMemsetAsync(out->d, 0, (size_t)grid->n_base * sizeof(double), ctx->stream));
    launch_line_survey(lines->field(0), lines->field(1), (int)lines->n, grid->range_min, grid->resolution, out->d,
                       grid->n_base, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_gather_compact_dev(lbl_ctx* ctx, lbl_buffer* gathered, int world_size, int64_t slot,
                                      const int64_t* first, const int64_t* count, lbl_buffer* out) try {
    if (!ctx || !first || !count) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    if (world_size < 1 || world_size > kMaxRanks) return fail(ctx, LBL_ERR_BAD_ARG, "world_size must be 1..%d", kMaxRanks);
    if (slot < 0) return fail(ctx, LBL_ERR_BAD_ARG, "negative slot");
    int rc;
    if ((rc = check_buf(ctx, gathered, (int64_t)world_size * slot, "gathered", true))) return rc;
    CompactArgs a;
    memset(&a, 0, sizeof a);
    long long need = 0, max_count = 0;
    for (int r = 0; r < world_size; ++r) {
        if (first[r] < 0 || count[r] < 0 || count[r] > slot) return fail(ctx, LBL_ERR_BAD_ARG, "rank %d: shard outside its slot", r);
        a.first[r] = first[r]; a.count[r] = count[r];
        need = std::max<long long>(need, first[r] + count[r]);
        max_count = std::max<long long>(max_count, count[r]);
    }
    if ((rc = check_buf(ctx, out, need, "out", true))) return rc;
    a.gathered = gathered->d; a.out = out->d; a.slot = slot; a.world = world_size;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    launch_gather_compact(a, max_count, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return LBL_OK;
} LBL_GUARD_END(ctx)

// ----------------------------------------------------------------------------------------
// graph capture of a launch sequence
// ----------------------------------------------------------------------------------------
struct lbl_graph {
    lbl_ctx* ctx;
    hipGraphExec_t exec;
    uint64_t epoch;
};

extern "C" int lbl_capture_begin(lbl_ctx* ctx) try {
    if (!ctx) return fail(nullptr, LBL_ERR_BAD_ARG, "ctx is NULL");
    if (ctx->capturing) return fail(ctx, LBL_ERR_STATE, "already capturing");
    bool chained = ctx->chain_pred != nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        for (lbl_ctx* c : g_contexts) chained = chained || c->chain_pred == ctx;
    }
    if (chained)
        return fail(ctx, LBL_ERR_STATE, "this context is part of an accumulate chain (lbl_ctx_chain_accumulate): the cross-context "
                                        "event waits cannot live in a captured graph; capture on an unchained context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_capture_end(lbl_ctx* ctx, lbl_graph** out) try {
    if (!ctx || !out) return fail(ctx, LBL_ERR_BAD_ARG, "NULL argument");
    *out = nullptr;
    if (!ctx->capturing) return fail(ctx, LBL_ERR_STATE, "not capturing");
    ctx->capturing = false;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (e != hipSuccess || !graph)
        return fail(ctx, LBL_ERR_HIP, "hipStreamEndCapture: %s (a call inside the capture was not capturable)", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(ctx, LBL_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    lbl_graph* g = new (std::nothrow) lbl_graph{ctx, exec, ctx->epoch};
    if (!g) { (void)hipGraphExecDestroy(exec); return fail(ctx, LBL_ERR_OOM, "host allocation failed"); }
    ctx->live_objects++;
    *out = g;
    return LBL_OK;
} LBL_GUARD_END(ctx)

extern "C" int lbl_graph_launch(lbl_graph* graph) try {
    if (!graph) return fail(nullptr, LBL_ERR_BAD_ARG, "graph is NULL");
    lbl_ctx* ctx = graph->ctx;
    if (ctx->capturing) return fail(ctx, LBL_ERR_STATE, "cannot launch a graph while capturing");
    if (graph->epoch != ctx->epoch)
        return fail(ctx, LBL_ERR_STATE, "graph is stale: a scratch buffer, schedule or descriptor it points at has changed since "
                                        "it was captured - capture it again");
    HIP_TRY(ctx, hipGraphLaunch(graph->exec, ctx->stream));
    return LBL_OK;
} LBL_GUARD_END(graph ? graph->ctx : nullptr)

extern "C" int lbl_graph_destroy(lbl_graph* graph) try {
    if (!graph) return LBL_OK;
    lbl_ctx* ctx = graph->ctx;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipGraphExecDestroy(graph->exec));
    ctx->live_objects--;
    delete graph;
    return LBL_OK;
} LBL_GUARD_END(graph ? graph->ctx : nullptr)

// hooks for lbl_comm.hip (kept out of the public header)
namespace lbl {
int comm_fail(lbl_ctx* ctx, int code, const char* msg) { return fail(ctx, code, "%s", msg); }
int ctx_device(lbl_ctx* ctx) { return ctx->device; }
bool ctx_capturing(lbl_ctx* ctx) { return ctx->capturing; }
lbl_ctx* buffer_ctx(lbl_buffer* buf) { return buf->ctx; }
}  // namespace lbl
