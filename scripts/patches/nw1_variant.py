#!/usr/bin/env python3
"""Experiment (round 5): one-wave workgroups for launches whose waves own their spans (LS = 1) - a workgroup of four waves
keeps its four wave slots until its slowest wave is done.  Patches a COPY of pyrad_amd/csrc.
usage: nw1_variant.py <dir with the csrc copy>"""
import sys
d = sys.argv[1]


def sub(path, pairs):
    t = open(path).read()
    for old, new in pairs:
        assert old in t, old[:60]
        t = t.replace(old, new, 1)
    open(path, "w").write(t)


sub(d + "/lbl_kernels.hip", [
    ("__global__ __launch_bounds__((LS > 4 ? 64 * LS : 256), (R >= 4 ? 4 : 1))                // HIP: min waves per SIMD",
     "__global__ __launch_bounds__((LS > 4 ? 64 * LS : (LS == 1 ? 64 : 256)), (R >= 4 ? 4 : 1))"),
    ("    constexpr int NW = LS > 4 ? LS : 4;              // wavefronts per workgroup (LS = 8: 512 threads)",
     "    constexpr int NW = LS > 4 ? LS : (LS == 1 ? 1 : 4);"),
    ("    return variant >= 3 ? 64 * R * ((LS > 4 ? LS : 4) / LS) : 256 * R;",
     "    return variant >= 3 ? 64 * R * ((LS > 4 ? LS : (LS == 1 ? 1 : 4)) / LS) : 256 * R;"),
    ("        default: hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 1, NT>), grid, dim3(256), pad, s, d_jobs, worklist); break;",
     "        default: hipLaunchKernelGGL((xsec_accumulate_lds_kernel<R, 1, NT>), grid, dim3(64), pad, s, d_jobs, worklist); break;"),
])
print("patched", d)
