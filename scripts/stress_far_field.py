"""Ad-hoc stress run (GPU): far-field kernel (variant 5) against the all-direct kernel (variant 3) on
random wide-window cells and launch shapes; prints the worst relative difference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err
from pyrad_amd import _native as nat, engine, synthetic

ctx = nat.Context(0)
first, count = int(sys.argv[1]), int(sys.argv[2])
# third argument "mid": every cell through the production shape (4 points per lane, unsplit spans) with the 32-point Gaussian
# runs forced - in exact mode the build whose series starts at 3 half-spans (round 6); dense line lists (up to 60,000 lines)
MID = len(sys.argv) > 3 and sys.argv[3] == "mid"
worst, top = 0.0, []
for seed in range(first, first + count):
    rng = np.random.default_rng(7000 + seed)
    base = float(rng.choice([0.0005, 0.001, 0.002]))
    P = float(np.exp(rng.uniform(np.log(100.0), np.log(30000.0))))
    T = int(rng.integers(180, 330))
    rmin = float(rng.choice([40.0, 650.0, 2300.0]))
    rmax = rmin + float(rng.uniform(8.0, 120.0))
    g = engine.layer_grid(P, rmin, rmax, base, False)
    n_lines = int(rng.integers(1, 60000 if MID else 6000))
    lines = synthetic.make_lines(8000 + seed, n_lines, g["eff_min"], g["eff_max"], decimals=7)
    lines["sw"] = 10.0 ** rng.uniform(-30.0, -18.0, n_lines)
    if rng.integers(0, 3) == 0:
        lines["gamma_air"] = lines["gamma_air"] * float(rng.uniform(1.0, 4.0))      # wide lines: Gaussian reach beyond the far threshold
    species = str(rng.choice(["co2", "h2o", "ch4", "o3"]))
    conc = float(rng.choice([4e-4, 1e-2, 0.3]))
    sp = synthetic.SPECIES[species]
    sel = engine.select_window(lines, g["eff_min"], g["eff_max"])
    iso = nat.IsoParams(float(T), float(P), conc, sp["molmass"], synthetic.q_value(species, T), sp["q296"])
    G = engine.native_grid(g)
    ctx.set_option("accum_variant", 3); ctx.set_option("accum_points_per_lane", 0); ctx.set_option("accum_line_split", 0)
    direct, c3 = ctx.xsec_accumulate(sel, iso, G)
    R = [0, 0, 1, 2, 4, 8][int(rng.integers(0, 6))]; LS = [0, 0, 1, 2, 4, 8][int(rng.integers(0, 6))]
    if MID:
        R, LS = 4, 1
    ctx.set_option("accum_gauss_run", 32 if MID else 0)
    ctx.set_option("accum_variant", 5); ctx.set_option("accum_points_per_lane", R); ctx.set_option("accum_line_split", LS)
    series, c5 = ctx.xsec_accumulate(sel, iso, G)
    assert tuple(c3) == tuple(c5) and np.all(np.isfinite(series))
    e = rel_err(series, direct)
    worst = max(worst, e)
    top.append((e, seed, g["W"], base, round(P, 1), n_lines, R, LS))
print("seeds %d..%d worst rel diff %.3e" % (first, first + count - 1, worst))
for t in sorted(top, reverse=True)[:5]:
    print("  %.2e seed %d W %d res %g P %s lines %d R %d LS %d" % t)
