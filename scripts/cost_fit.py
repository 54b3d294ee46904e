"""Diagnostic (needs scripts/bin/libpyrad_hip_dbg.so): fit the host's per-span cost model to measured
wave durations of the far-field kernel on C2 in positional order (workgroup b = tile b)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine, synthetic
sys.argv = ["x"]
import bench
ctx = nat.Context(0)
ctx.set_option("accum_longest_first", 0)
ctx.set_option("accum_tile_order", 1)
R, LS = 4, 2
ctx.set_option("accum_points_per_lane", R); ctx.set_option("accum_line_split", LS)
cfg, _ = bench.build_workload("C2", 1)
L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                         cfg["base_resolution"], False, keep_host_lines=True)
for _ in range(3):
    L.enqueue_xsec()
ctx.sync()
tile_pts = 64 * R * (4 // LS)
nb = 8 * ((L.n + tile_pts - 1) // tile_pts + 7) // 8
n = nb * 4 * 3
buf = (C.c_uint64 * n)()
ctx.lib.lbl_debug_times.restype = C.c_int
ctx.lib.lbl_debug_times.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
ctx.lib.lbl_debug_times(ctx.h, buf, n)
t = np.array(buf[:n], dtype=np.uint64).reshape(-1, 4, 3)
dur = (t[:, :, 1].astype(np.float64) - t[:, :, 0].astype(np.float64)) / 100.0      # us, [workgroup, wave]
sel = L._keep[0]
idx = ((sel["nu"] - cfg["range_min"]) / cfg["base_resolution"]).astype(np.int64)
H = L.g["W"] - 2
span = 64 * R
reach = 4 * 32 * R
below = lambda v: np.searchsorted(idx, v, side="left")
rows, y = [], []
n_spans = (L.n + span - 1) // span
for s in range(n_spans):
    lo = s * span; hi = min(lo + span - 1, L.n - 1)
    iA, iD, iB, iC = below(lo - H), below(hi + H + 1), below(hi - H), below(lo + H + 1)
    iF1 = min(max(below(lo + 32 * R - reach), iB), iC); iF2 = min(max(below(lo + 32 * R + reach), iF1), iC)
    wg, grp = divmod(s, 4 // LS)
    d = dur[wg, grp * LS:(grp + 1) * LS]
    if wg >= dur.shape[0] or d.min() <= 0:
        continue
    rows.append([iF2 - iF1, (iB - iA) + (iD - iC), (iF1 - iB) + (iC - iF2), 1.0])
    y.append(d.mean())
X = np.array(rows, dtype=np.float64); y = np.array(y)
w, *_ = np.linalg.lstsq(X, y, rcond=None)
print("spans fitted", len(y), "duration us min/median/max", y.min().round(1), np.median(y).round(1), y.max().round(1))
print("least squares  us per: near %.4f  edge %.4f  far %.5f  const %.2f" % tuple(w))
print("relative to near: edge %.2f far %.4f const %.1f near-lines" % (w[1] / w[0], w[2] / w[0], w[3] / w[0]))
pred = X @ w
print("fit residual: rms %.2f us, max |err| %.2f us, corr %.4f" % (np.sqrt(np.mean((pred - y) ** 2)), np.abs(pred - y).max(), np.corrcoef(pred, y)[0, 1]))
cur = X @ np.array([5.0 * R + 29.0, 8.0 * R, (3 * 30 + 12) / 64.0, 600.0])
print("current model corr %.4f; ratio measured/model percentiles" % np.corrcoef(cur, y)[0, 1], np.percentile(y / cur * np.median(cur) / np.median(y), [0, 10, 50, 90, 100]).round(3))
