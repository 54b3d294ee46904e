"""Diagnostic (needs the instrumented build scripts/bin/libpyrad_hip_phase.so, scripts/make_phase_lib.sh): per-wave
start/end realtime and placement (XCD, CU, SIMD) of the far-field accumulate kernel; scripts/phase_times.py splits the
same stamps by phase.
    WORKLOAD=C3 SHARD=8,4 PYRAD_HIP_LIB=$PWD/scripts/bin/libpyrad_hip_phase.so python scripts/wave_times.py"""
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
workload = os.environ.get("WORKLOAD", "C2")        # WORKLOAD=C3 SHARD=8,4: one shard of 8 of the mixed cell
cfg, _ = bench.build_workload(workload, scale)
shard = None
if os.environ.get("SHARD"):
    G, r = (int(v) for v in os.environ["SHARD"].split(","))
    shard = (G, r)
L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                         cfg["base_resolution"], False, shard=shard)
for _ in range(3):
    L.enqueue_xsec()
ctx.sync()
pts = L.count * len(L.jobs)
nb = 8 * ((pts // (64 * int(os.environ.get("R", "4")) * (4 // int(os.environ.get("LS", "1" if workload != "C2" else "2")))) + 8) // 8)
n = nb * 4 * 6
buf = (C.c_uint64 * n)()
ctx.lib.lbl_debug_times.restype = C.c_int
ctx.lib.lbl_debug_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.lbl_debug_times(ctx.h, buf, n)
t = np.array(buf[:n], dtype=np.uint64).reshape(-1, 6)[:, [0, 4, 5]]       # entry, exit, placement (the phase stamps between)
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

# dispatch: does every tier of n_cu consecutive workgroups (worklist order) land on distinct CUs?
wg_cu = cuid.reshape(-1, 4)[:, 0] if len(cuid) % 4 == 0 else None
if wg_cu is not None:
    ncu = len(u)
    tiers = [wg_cu[i:i + ncu] for i in range(0, len(wg_cu), ncu)]
    print("tiers of", ncu, "workgroups -> distinct CUs per tier:", [len(set(t.tolist())) for t in tiers])
    per_cu_wgs = np.array([np.sum(wg_cu == x) for x in u])
    print("workgroups per CU min/median/max", per_cu_wgs.min(), int(np.median(per_cu_wgs)), per_cu_wgs.max())
busy = np.array([np.sum((end - start)[cuid == x]) for x in u]) / 4.0     # wave-us per SIMD of the CU
print("per-CU resident wave time / 4 SIMDs: percentiles", np.percentile(busy, [0, 10, 50, 90, 100]).round(1))
# SIMD level: how many waves does each SIMD hold, and when does it finish?
sid = cuid * 4 + simd
us, cs = np.unique(sid, return_counts=True)
print("SIMDs used", len(us), "waves per SIMD histogram", dict(zip(*np.unique(cs, return_counts=True))))
fin = np.array([end[sid == x].max() for x in us])
print("per-SIMD finish time percentiles", np.percentile(fin, [0, 10, 50, 90, 100]).round(1))
for k in sorted(set(cs.tolist())):
    f = fin[cs == k]
    print("  SIMDs with %d waves: %4d, finish median %.1f max %.1f" % (k, len(f), np.median(f), f.max()))
work = end - start
tot = np.array([work[sid == x].sum() for x in us])
print("sum of wave durations per SIMD percentiles", np.percentile(tot, [0, 10, 50, 90, 100]).round(1))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print("XCC %d: waves %4d, wave duration median %.1f, finish median %.1f max %.1f, sum of durations %.0f"
          % (x, m.sum(), np.median(work[m]), np.median(end[m]), end[m].max(), work[m].sum()))
