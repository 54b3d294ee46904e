"""Diagnostic (needs the instrumented build scripts/bin/libpyrad_hip_dbg.so): per-wave start/end
realtime and placement of the LS accumulate kernel on C2."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine, synthetic
sys.argv = ["x"]
import bench
scale = int(os.environ.get("SCALE", "1"))
ctx = nat.Context(0)
for k in ("R", "LS"):
    if os.environ.get(k):
        ctx.set_option({"R": "accum_points_per_lane", "LS": "accum_line_split"}[k], int(os.environ[k]))
cfg, _ = bench.build_workload("C2", scale)
L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                         cfg["base_resolution"], False)
for _ in range(3):
    L.enqueue_xsec()
ctx.sync()
nb = 8 * ((L.n // (64 * int(os.environ.get("R", "2")) * (4 // int(os.environ.get("LS", "4")))) + 8) // 8)
n = nb * 4 * 3
buf = (C.c_uint64 * n)()
ctx.lib.lbl_debug_times.restype = C.c_int
ctx.lib.lbl_debug_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.lbl_debug_times(ctx.h, buf, n)
t = np.array(buf[:n], dtype=np.uint64).reshape(-1, 3)
t = t[t[:, 1] > 0]
t0 = float(t[:, 0].min())
start = (t[:, 0].astype(np.float64) - t0) / 100.0
end = (t[:, 1].astype(np.float64) - t0) / 100.0
hw = t[:, 2]
xcc = (hw >> np.uint64(32)).astype(int)
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(int)
cu = (hwid >> 8) & 0xF
se = (hwid >> 13) & 0x7
simd = (hwid >> 4) & 0x3
cuid = xcc * 1000 + se * 16 + cu
print("waves", len(t), "kernel span %.1f us" % end.max())
print("start us percentiles", np.percentile(start, [0, 10, 50, 90, 100]).round(1))
print("end   us percentiles", np.percentile(end, [0, 10, 50, 90, 100]).round(1))
print("duration percentiles", np.percentile(end - start, [0, 10, 50, 90, 100]).round(1))
# active waves over time
for tt in np.linspace(0, end.max(), 11):
    act = ((start <= tt) & (end > tt)).sum()
    print("t=%6.1f us active waves %5d" % (tt, act))
u, c = np.unique(cuid, return_counts=True)
print("distinct CUs", len(u), "waves per CU min/median/max", c.min(), int(np.median(c)), c.max())
last = np.array([end[cuid == x].max() for x in u])
print("per-CU finish time percentiles", np.percentile(last, [0, 10, 50, 90, 100]).round(1))
