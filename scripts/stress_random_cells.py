"""Ad-hoc stress run (GPU): the random-cell differential test of tests/test_gpu_parity.py over many
more seeds than the suite carries, plus random launch shapes.  Prints the worst relative error."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err, point_tolerance, rel_err_points
from pyrad_amd import _native as nat, engine, synthetic
from oracle import pyrad_oracle as orc

ctx = nat.Context(0)
first, count = int(sys.argv[1]), int(sys.argv[2])
worst, fails, top = 0.0, [], []
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    base = float(rng.choice([0.01, 0.001, 0.0001]))
    dyn = bool(rng.integers(0, 2))
    P = float(np.exp(rng.uniform(np.log(0.05), np.log(20000.0))))
    T = int(rng.integers(150, 351))
    rmin = float(rng.choice([0.0, 0.5, 37.0, 600.0, 2499.3, 12000.0]))
    g0 = orc.layer_grid(P, rmin, rmin + 1.0, base, dyn)
    width = float(min(rng.uniform(0.02, 30.0), 60000 * g0["resolution"], 20000 * base))
    rmax = rmin + width
    g = orc.layer_grid(P, rmin, rmax, base, dyn)
    if g["W"] < 1 or g["n_base"] < 1 or g["n_work"] < 1:
        continue
    n_lines = int(rng.choice([0, 1, 2, 17, 150, 400, 1500]))
    n_lines = int(min(n_lines, max(1, 2e6 // max(g["W"], 1)))) if n_lines else 0
    lo, hi = g["eff_min"], g["eff_max"]
    lines = synthetic.make_lines(5000 + seed, n_lines, lo, hi, decimals=7) if n_lines else {k: np.zeros(0) for k in synthetic.FIELDS}
    species = str(rng.choice(["co2", "h2o", "ch4", "o3"]))
    conc = float(rng.choice([4e-4, 1e-2, 0.5, 1.8e-6]))
    R = [0, 0, 1, 2, 4, 8][int(rng.integers(0, 6))]
    LS = [0, 0, 1, 2, 4, 8][int(rng.integers(0, 6))]
    ctx.set_option("accum_points_per_lane", R); ctx.set_option("accum_line_split", LS)
    sp = synthetic.SPECIES[species]
    sel = engine.select_window(lines, lo, hi)
    iso = nat.IsoParams(float(T), float(P), conc, sp["molmass"], synthetic.q_value(species, T), sp["q296"])
    xs, counts = ctx.xsec_accumulate(sel, iso, engine.native_grid(g))
    ref, rc = orc.create_cross_section(orc.select_window(lines, lo, hi), T, P, conc, sp["molmass"],
                                       synthetic.q_value(species, T), sp["q296"], g)
    try:
        assert tuple(counts) == tuple(rc)
        e = rel_err(xs, ref)
    except AssertionError as ex:
        fails.append((seed, str(ex)[:80])); continue
    worst = max(worst, e)
    top.append((e, seed, g['W'], g['resolution'], round(P, 2), n_lines, R, LS, species))
    # the suite's per-point bound (conftest.point_tolerance: 2e-12 + the nu -> 0 amplification of the reference's own
    # cancellation in 1 - exp(-c2 nu / T); cells that start at 0 cm^-1 exceed any flat figure)
    tol = point_tolerance(orc.x_axis(rmin, rmax, base), T, g["dfc"])
    floor = float(np.max(np.abs(ref))) * 1e-250 if ref.size else 0.0
    ep = rel_err_points(xs, ref, floor)
    if ref.size and not np.all(ep <= tol):
        fails.append((seed, float(np.max(ep / tol)), g["W"], R, LS))
print("seeds %d..%d worst rel err %.3e fails %s" % (first, first + count - 1, worst, fails))
for t in sorted(top, reverse=True)[:8]:
    print("  err %.2e seed %d W %d res %g P %s lines %d R %d LS %d %s" % t)
