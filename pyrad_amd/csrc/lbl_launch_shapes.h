// Launch-shape arithmetic of the line-by-line engine that needs no device: sizes of dispatch lists and scratch blocks, what the
// device-side schedule build covers, tile and block counts.  Shared by lbl_kernels.hip (the launchers) and by the CPU sanitizer
// harness of the host shim (tests/host_shim/: lbl_api.hip compiled as plain C++ against a stand-in HIP runtime, with launchers
// that touch exactly the ranges the kernels touch), so that AddressSanitizer checks the shim's buffer sizing against the SAME
// formulas the kernels are launched with.
#pragma once
#include <cstddef>

namespace lbl {

// Entries of a launch's dispatch list: the tiles, or - single round on a chip of 8 XCDs - eight runs of `m_cap` positions
inline int sched_xcd_positions(int total_tiles, int n_cu) {
    if (total_tiles <= 0 || total_tiles > 4 * n_cu || total_tiles > 1024 || n_cu % 8 != 0 || n_cu < 64 || n_cu > 512) return 0;
    const int b = n_cu / 8;                                      // bins per XCD; up to twice an eighth of the tiles per XCD (a
    int m = 2 * ((total_tiles + 7) / 8);                         // sparse spectral region is many cheap tiles), in whole tiers
    m = (m + b - 1) / b * b;                                     // of b positions, at most 7
    if (m > 7 * b) m = 7 * b;
    if (8 * m < total_tiles) return 0;
    return m;
}
inline int sched_launch_items(int total_tiles, int n_cu, bool xcd_pack) {
    const int m = xcd_pack ? sched_xcd_positions(total_tiles, n_cu) : 0;
    return m > 0 ? 8 * m : total_tiles;
}

// What the device build covers (else the caller builds the schedule on the host), and the scratch it needs.
inline bool sched_device_supported(int total_tiles, int n_cu) {
    if (total_tiles <= 4 * n_cu) return total_tiles <= 1024 && n_cu <= 512;
    return total_tiles <= (1 << 20);
}
inline int sched_key_stride(int total_tiles) {
    int p = 1;
    while (p < total_tiles) p <<= 1;
    return p;
}
inline size_t sched_scratch_bytes(int total_tiles) {         // tile costs | items | prefix | global sort keys (8 parts)
    const size_t n = (size_t)total_tiles;
    return ((n * 4 + 255) & ~(size_t)255) + ((n * 8 + 255) & ~(size_t)255) + (((n + 1) * 8 + 255) & ~(size_t)255) +
           8 * (size_t)sched_key_stride(total_tiles) * 8 + 256;
}


// partial sums of the band integral: 16384 points per block
inline int band_partial_count(long long n) {
    long long b = (n + 16383) / 16384;
    return (int)(b < 1 ? 1 : b);
}

// grid points one workgroup of the accumulate kernels covers
inline int accumulate_tile_points(int R, int LS, int variant) {
    return variant >= 3 ? 64 * R * ((LS > 4 ? LS : 4) / LS) : 256 * R;
}

}  // namespace lbl
