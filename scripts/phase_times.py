"""Diagnostic (needs scripts/bin/libpyrad_hip_phase.so): where a wave of the far-field accumulate kernel spends its
time - edge lines (skewed walk), series phase, near lines + Gaussian runs, output stage - per occupancy of its SIMD.
    WORKLOAD=C3 SHARD=8,4 PYRAD_HIP_LIB=$PWD/scripts/bin/libpyrad_hip_phase.so python scripts/phase_times.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine
sys.argv = ["x"]
import bench
ctx = nat.Context(0)
if os.environ.get("ACCURACY"):
    ctx.set_option("accuracy", int(os.environ["ACCURACY"]))
workload = os.environ.get("WORKLOAD", "C3")
cfg, _ = bench.build_workload(workload, 1)
shard = tuple(int(v) for v in os.environ["SHARD"].split(",")) if os.environ.get("SHARD") else None
L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                         cfg["base_resolution"], False, shard=shard)
for _ in range(3):
    L.enqueue_xsec()
ctx.sync()
n_wg = min(65536, (L.count * len(L.jobs) + 1023) // 1024)
n = n_wg * 4 * 6
buf = (C.c_uint64 * n)()
ctx.lib.lbl_debug_times.restype = C.c_int
ctx.lib.lbl_debug_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.lbl_debug_times(ctx.h, buf, n)
t = np.array(buf[:n], dtype=np.uint64).reshape(-1, 6)
t = t[t[:, 4] > 0]
t0 = float(t[:, 0].min())
st = (t[:, :5].astype(np.float64) - t0) / 100.0            # microseconds (100 MHz counter)
hw = t[:, 5]
xcc = (hw >> np.uint64(32)).astype(int)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(int)
sid = (xcc * 1000 + ((hwid >> 13) & 0x7) * 16 + ((hwid >> 8) & 0xF)) * 4 + ((hwid >> 4) & 0x3)
print("waves", len(t), "kernel span %.1f us" % st[:, 4].max())
names = ["edges (skewed walk)", "series phase", "near lines + Gaussian runs", "output stage"]
dur = np.diff(st, axis=1)
us, inv, cs = np.unique(sid, return_inverse=True, return_counts=True)
occ = cs[inv]
for k in sorted(set(occ.tolist())):
    m = occ == k
    print("waves on SIMDs holding %d waves: %d; whole wave median %.1f us; phases (median us): %s" % (
        k, m.sum(), np.median(st[m, 4] - st[m, 0]), ", ".join("%s %.1f" % (nm, np.median(dur[m, i])) for i, nm in enumerate(names))))
print("all waves: phase medians", {nm: round(float(np.median(dur[:, i])), 2) for i, nm in enumerate(names)},
      "sums of medians %.1f" % sum(np.median(dur[:, i]) for i in range(4)))
print("phase p10 / p90:", {nm: (round(float(np.percentile(dur[:, i], 10)), 1), round(float(np.percentile(dur[:, i], 90)), 1)) for i, nm in enumerate(names)})
