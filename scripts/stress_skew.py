"""Ad-hoc stress run (GPU) of the round-3 kernels over many random cells, whole spectra against the oracle with the
per-point tolerance of tests/conftest.py:
  mode "skew":  every job through the skewed-range kernel (accum_skew 2), random points per lane
  mode "edges": wide windows (>= 642 points) through the far-field kernel at R = 4, unsplit spans (skew_edges)
usage: stress_skew.py <mode> <first seed> <count>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import point_tolerance, rel_err_points
from pyrad_amd import _native as nat, engine, synthetic
from oracle import pyrad_oracle as orc

mode, first, count = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ctx = nat.Context(0)
worst, fails, done = 0.0, [], 0
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    base = float(rng.choice([0.01, 0.001, 0.0001]))
    if mode == "edges":
        base = float(rng.choice([0.001, 0.0001]))
        P = float(np.exp(rng.uniform(np.log(140.0 * base / 0.001), np.log(3000.0 * base / 0.001))))
        dyn = False
    else:
        P = float(np.exp(rng.uniform(np.log(0.05), np.log(20000.0))))
        dyn = bool(rng.integers(0, 2))
    T = int(rng.integers(150, 351))
    rmin = float(rng.choice([0.0, 0.5, 37.0, 600.0, 2499.3, 12000.0]))
    g0 = orc.layer_grid(P, rmin, rmin + 1.0, base, dyn)
    width = float(min(rng.uniform(0.02, 30.0), 60000 * g0["resolution"], 20000 * base))
    rmax = rmin + width
    g = orc.layer_grid(P, rmin, rmax, base, dyn)
    if g["W"] < 1 or g["n_base"] < 1 or g["n_work"] < 1 or (mode == "edges" and g["W"] < 642):
        continue
    n_lines = int(rng.choice([1, 2, 17, 150, 400, 1500, 4000]))
    n_lines = int(min(n_lines, max(1, 4e6 // max(g["W"], 1))))
    try:
        lines = synthetic.make_lines(6000 + seed, n_lines, g["eff_min"], g["eff_max"], decimals=int(rng.choice([3, 5, 7])))
    except RuntimeError:           # too many lines for that few decimals in this window
        lines = synthetic.make_lines(6000 + seed, n_lines, g["eff_min"], g["eff_max"], decimals=9)
    species = str(rng.choice(["co2", "h2o", "ch4", "o3"]))
    conc = float(rng.choice([4e-4, 1e-2, 0.5, 1.8e-6]))
    sp = synthetic.SPECIES[species]
    sel = engine.select_window(lines, g["eff_min"], g["eff_max"])
    iso = nat.IsoParams(float(T), float(P), conc, sp["molmass"], synthetic.q_value(species, T), sp["q296"])
    if mode == "skew":
        ctx.set_option("accum_skew", 2)
        ctx.set_option("accum_skew_points_per_lane", int(rng.choice([1, 2, 4, 8])))
    else:
        ctx.set_option("accum_points_per_lane", 4)
        ctx.set_option("accum_line_split", 1)
    xs, counts = ctx.xsec_accumulate(sel, iso, engine.native_grid(g))
    ref, rc = orc.create_cross_section(sel, T, P, conc, sp["molmass"], synthetic.q_value(species, T), sp["q296"], g)
    tol = point_tolerance(orc.x_axis(rmin, rmax, base), T, g["dfc"])
    floor = float(np.max(np.abs(ref))) * 1e-250 if ref.size else 0.0
    e = rel_err_points(xs, ref, floor)
    done += 1
    worst = max(worst, float((e / tol).max()))
    if tuple(counts) != tuple(rc) or not np.all(e <= tol):
        fails.append((seed, float(e.max()), g["W"], n_lines))
print("%s: seeds %d..%d, %d cells, worst error / tolerance %.3g, fails %s" % (mode, first, first + count - 1, done, worst, fails))
