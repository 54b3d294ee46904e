#!/usr/bin/env python3
"""Experiment (round 5, measured and NOT adopted; DESIGN.md §4 "Round 5"): the far-field loop's per-chunk decisions on the
scalar unit.  Patches a COPY of pyrad_amd/csrc: K1 writes the largest Gaussian reach per block of 256 records, the loop reads
the chunk's end centre indices and block reaches with s_load_dword and decides term count and Gaussian test in scalar code.
usage: reach_variant.py <dir with the csrc copy>"""
import sys
d = sys.argv[1]


def sub(path, pairs):
    t = open(path).read()
    for old, new in pairs:
        assert old in t, old[:60]
        t = t.replace(old, new, 1)
    open(path, "w").write(t)


sub(d + "/lbl_device.h", [
    ("    const int32_t* span_tab;\n    // Fused layer step of a single-line-list layer",
     "    const int32_t* span_tab;\n    const int32_t* reach;\n    // Fused layer step of a single-line-list layer"),
    ("    unsigned int* block_counts;           // [blocks of 256 lines][3]: per-block regime counts, no atomics\n",
     "    unsigned int* block_counts;           // [blocks of 256 lines][3]: per-block regime counts, no atomics\n    int32_t* reach;\n"),
    ("    int32_t n_total, blocks;                         // lines of all its lists; ceil(n_total / 256)\n};",
     "    int32_t n_total, blocks;                         // lines of all its lists; ceil(n_total / 256)\n    int32_t* reach;\n};"),
])
sub(d + "/lbl_kernels.hip", [
    ("__global__ __launch_bounds__(256) void line_prep_kernel(const PrepJob* __restrict__ jobs) {",
     "__device__ __forceinline__ int wave_max_shfl_i32(int v) {\n#pragma unroll\n    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d));\n    return v;\n}\n\n"
     "__global__ __launch_bounds__(256) void line_prep_kernel(const PrepJob* __restrict__ jobs) {"),
    ("    int regime = -1;\n    if (i < J.n_lines) {\n        HotRec r;\n        ColdRec rc;\n        long long idx;\n        prep_one_line(J, i, r, rc, idx, regime);\n        J.hot[i] = r;\n        J.cold[i] = rc;\n        J.cidx[i] = (int32_t)idx;\n    }",
     "    int regime = -1, reach = 0;\n    if (i < J.n_lines) {\n        HotRec r;\n        ColdRec rc;\n        long long idx;\n        prep_one_line(J, i, r, rc, idx, regime);\n        J.hot[i] = r;\n        J.cold[i] = rc;\n        J.cidx[i] = (int32_t)idx;\n        reach = r.dgi;\n    }"),
    ("    __shared__ unsigned int s_cnt[4][3];\n    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n    for (int k = 0; k < 3; ++k) {\n        const unsigned long long m = __ballot(regime == k);\n        if (lane == 0) s_cnt[wave][k] = (unsigned int)__popcll(m);\n    }\n    __syncthreads();\n    if (threadIdx.x < 3)\n        J.block_counts[blockIdx.x * 3 + threadIdx.x] =\n            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];\n}",
     "    __shared__ unsigned int s_cnt[4][3];\n    __shared__ int s_reach[4];\n    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;\n    for (int k = 0; k < 3; ++k) {\n        const unsigned long long m = __ballot(regime == k);\n        if (lane == 0) s_cnt[wave][k] = (unsigned int)__popcll(m);\n    }\n    const int wave_reach = wave_max_shfl_i32(reach);\n    if (lane == 0) s_reach[wave] = wave_reach;\n    __syncthreads();\n    if (threadIdx.x < 3)\n        J.block_counts[blockIdx.x * 3 + threadIdx.x] =\n            s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] + s_cnt[3][threadIdx.x];\n    if (threadIdx.x == 0 && J.reach) J.reach[blockIdx.x] = max(max(s_reach[0], s_reach[1]), max(s_reach[2], s_reach[3]));\n}"),
    ("    int regime = -1, list = -1;\n    if (t < M.n_total) {", "    int regime = -1, list = -1, reach = 0;\n    if (t < M.n_total) {"),
    ("        M.cidx[t] = (int32_t)idx;\n    }\n    // regime counters per list and block of 256 merged positions\n    __shared__ unsigned int s_cnt[4][kMaxIso][3];\n",
     "        M.cidx[t] = (int32_t)idx;\n        reach = r.dgi;\n    }\n    __shared__ unsigned int s_cnt[4][kMaxIso][3];\n    __shared__ int s_reach[4];\n    { const int wr = wave_max_shfl_i32(reach); if ((threadIdx.x & 63) == 0) s_reach[threadIdx.x >> 6] = wr; }\n"),
    ("    __syncthreads();\n    for (int q = threadIdx.x; q < M.n_lists * 3; q += blockDim.x) {",
     "    __syncthreads();\n    if (threadIdx.x == 0 && M.reach) M.reach[blockIdx.x] = max(max(s_reach[0], s_reach[1]), max(s_reach[2], s_reach[3]));\n    for (int q = threadIdx.x; q < M.n_lists * 3; q += blockDim.x) {"),
    ("__device__ __forceinline__ void far_field_lines(const HotRec* hot, const ColdRec* cold, int m0, int m1, int stride,",
     "__device__ __forceinline__ void far_field_lines(const HotRec* hot, const ColdRec* cold, const int32_t* __restrict__ cidx,\n                                                const int32_t* __restrict__ reach, int m0, int m1, int stride,"),
    ("    for (int c0 = m0; c0 < m1; c0 += stride) {\n        const bool valid = c0 + lane < m1;\n        const v2f64 w0 = h0, w1 = h1;\n        if (c0 + stride + lane < m1) {\n            const long long r = (long long)(c0 + stride + lane) * 2;\n            h0 = gh[r]; h1 = gh[r + 1];\n        }\n        const int ci = (int)w0.x;\n        const int dgi = __double2loint(w1.y), fl = __double2hiint(w1.y);\n        const bool gauss = valid && max(0, max(ci - whi, wlo - ci)) < dgi;\n        const unsigned long long gmask = __ballot(gauss);\n",
     "    typedef const int32_t __attribute__((address_space(4)))* ScalarI32;\n    const ScalarI32 s_cidx = (ScalarI32)(unsigned long long)cidx;\n    const ScalarI32 s_reach = (ScalarI32)(unsigned long long)reach;\n    auto ends = [&](int c, int& ca, int& cb, int& rr) {\n        const int last = min(c + 63, m1 - 1);\n        ca = s_cidx[c]; cb = s_cidx[last];\n        rr = reach ? max(s_reach[c >> 8], s_reach[last >> 8]) : 0x7fffffff;\n    };\n    int n_ca = 0, n_cb = 0, n_rr = 0;\n    if (m0 < m1) ends(m0, n_ca, n_cb, n_rr);\n    for (int c0 = m0; c0 < m1; c0 += stride) {\n        const bool valid = c0 + lane < m1;\n        const v2f64 w0 = h0, w1 = h1;\n        const int ca = n_ca, cb = n_cb, rr = n_rr;\n        if (c0 + stride + lane < m1) {\n            const long long r = (long long)(c0 + stride + lane) * 2;\n            h0 = gh[r]; h1 = gh[r + 1];\n        }\n        if (c0 + stride < m1) ends(c0 + stride, n_ca, n_cb, n_rr);\n        const int off_a = max(0, max(ca - whi, wlo - ca)), off_b = max(0, max(cb - whi, wlo - cb));\n        unsigned long long gmask = 0ull;\n        int fl = 0;\n        bool gauss = false;\n        if (rr > min(off_a, off_b)) {\n            const int ci = (int)w0.x;\n            const int dgi = __double2loint(w1.y);\n            fl = __double2hiint(w1.y);\n            gauss = valid && max(0, max(ci - whi, wlo - ci)) < dgi;\n            gmask = __ballot(gauss);\n        }\n"),
    ("        const int nv = min(64, m1 - c0);\n        const double dmin = fmin(fabs(readlane_f64(w0.x, 0) - xc), fabs(readlane_f64(w0.x, nv - 1) - xc));\n",
     "        const int dmin2 = min(abs(2 * (ca - wlo) - (64 * R - 1)), abs(2 * (cb - wlo) - (64 * R - 1)));\n"),
    ("        if (dmin < 32.0 * hh) {", "        if (dmin2 < 2 * 32 * 32 * R) {"),
    ("            if (dmin < 16.0 * hh) {", "            if (dmin2 < 2 * 16 * 32 * R) {"),
    ("                if (dmin < 8.0 * hh) {", "                if (dmin2 < 2 * 8 * 32 * R) {"),
    ("                    if (FT::t4 < NT && dmin < 4.0 * hh) series_terms", "                    if (FT::t4 < NT && dmin2 < 2 * 4 * 32 * R) series_terms"),
    ("far_field_lines<R, NTC>(J.hot, J.cold, iB + ((part + 1) % LS) * 64, iF1, 64 * LS,", "far_field_lines<R, NTC>(J.hot, J.cold, J.cidx, J.reach, iB + ((part + 1) % LS) * 64, iF1, 64 * LS,"),
    ("far_field_lines<R, NTC>(J.hot, J.cold, iF2 + ((part + 2) % LS) * 64, iC, 64 * LS,", "far_field_lines<R, NTC>(J.hot, J.cold, J.cidx, J.reach, iF2 + ((part + 2) % LS) * 64, iC, 64 * LS,"),
])
sub(d + "/lbl_api.hip", [
    ("    DeviceArena merge_tmp;   // merged layer jobs:", "    DeviceArena reach;\n    DeviceArena merge_tmp;   // merged layer jobs:"),
    ("&ctx->zeros, &ctx->sched, &ctx->merge_tmp, &ctx->ktmp};", "&ctx->zeros, &ctx->sched, &ctx->merge_tmp, &ctx->ktmp, &ctx->reach};"),
    ("    if ((rc = arena_reserve(ctx, ctx->counts, cnt_bytes))) return rc;\n",
     "    if ((rc = arena_reserve(ctx, ctx->counts, cnt_bytes))) return rc;\n    std::vector<size_t> reach_off(n_jobs);\n    size_t reach_blocks = 0;\n    for (int j = 0; j < n_jobs; ++j) { reach_off[j] = reach_blocks; reach_blocks += (job_lines[j] + 255) / 256 + 1; }\n    if ((rc = arena_reserve(ctx, ctx->reach, std::max<size_t>(reach_blocks, 1) * sizeof(int32_t)))) return rc;\n"),
    ("            m.n_total = (int32_t)job_lines[j]; m.blocks = (int32_t)((job_lines[j] + 255) / 256);\n",
     "            m.n_total = (int32_t)job_lines[j]; m.blocks = (int32_t)((job_lines[j] + 255) / 256);\n            m.reach = (int32_t*)ctx->reach.ptr + reach_off[j];\n"),
    ("            p.block_counts = d_counts + (size_t)l * blocks_per_job * 3;\n",
     "            p.block_counts = d_counts + (size_t)l * blocks_per_job * 3;\n            p.reach = scattered ? nullptr : (int32_t*)ctx->reach.ptr + reach_off[j];\n"),
    ("            a.span_tab = g.tabs ? g.tabs + g.tab_off[(size_t)(k - g.first)] : nullptr;\n",
     "            a.span_tab = g.tabs ? g.tabs + g.tab_off[(size_t)(k - g.first)] : nullptr;\n            a.reach = (const int32_t*)ctx->reach.ptr + reach_off[j];\n"),
])
print("patched", d)
