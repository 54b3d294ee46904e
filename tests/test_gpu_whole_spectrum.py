"""GPU: EVERY grid point of the BASELINE configurations at full size against the plain-C oracle.

The sampled tests of test_gpu_fullsize.py look at a few dozen points; the accumulate kernel decides
its code path per span (edge / near / far / Gaussian-run classes, line split, worklist order), so a
mis-classified span can hide between samples.  oracle/lbl_oracle.c restates the reference's loop
(pyradClasses.py:361-407, pyradLineshape.py, pyradIntensity.py:16-32) at 2e8 evals/s per host core:
all of C2 in 3 s, each of C3's three line lists in ~6 s (one thread per list: ctypes releases the GIL,
no process is forked from the GPU process).  It is pinned to the goldens by tests/test_oracle_golden.py.

Tolerance: conftest.point_tolerance (per point: 2e-12 + the nu -> 0 amplification of the
stimulated-emission factor stated in ulps); the worst point of every comparison is printed.

Every test runs in both accuracy modes of the library (lbl_set_option "accuracy"): "exact" (the default; tolerance as
above, measured 1e-14) and "budget" (18..7 far-field series terms by distance, Gaussian cut-off at 2^-34 of the line's
Lorentz term: stated bound 1e-9 relative on the absorption coefficient, which replaces the 2e-12 of the per-point
tolerance; BASELINE north_star asks for 1e-6).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import point_tolerance, rel_err_points
from pyrad_amd import synthetic

pytestmark = pytest.mark.gpu
K_BOLTZMANN = 1.38064852E-23


@pytest.fixture(scope="module")
def ctx():
    from pyrad_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


MODES = {"exact": (0, None), "budget": (1, 1e-9)}


def tolerance(mode, xa, T, dfc):
    return point_tolerance(xa, T, dfc) if MODES[mode][1] is None else point_tolerance(xa, T, dfc, rtol_base=MODES[mode][1])


def oracle_xsecs(jobs):
    """jobs: list of (lines, T, P, conc, molmass, q_T, q296, grid) -> [(xsec, regime counts, evals)], one
    host thread per job"""
    from oracle import c_oracle
    c_oracle.load()
    with ThreadPoolExecutor(max_workers=min(len(jobs), 12)) as ex:
        return list(ex.map(lambda j: c_oracle.create_cross_section_work(*j), jobs))


def worst(tag, got, ref, tol):
    e = rel_err_points(got, ref)
    i = int(np.argmax(e / tol))
    print("%s: max rel err %.3e (point %d of %d, tolerance there %.1e)" % (tag, float(e.max()), i, e.size, tol[i]))
    assert np.all(e <= tol), (tag, float(e.max()), i)
    return float(e.max())


def mols_of(cfg):
    from pyrad_amd.model import concentration_from_kwargs
    out = []
    for mol in cfg["molecules"]:
        sp = synthetic.SPECIES[mol["species"]]
        out.append(dict(conc=concentration_from_kwargs(**mol["conc"]), species=mol["species"],
                        isotopologues=[dict(lines=mol["lines"], molmass=sp["molmass"],
                                            q_T=synthetic.q_value(mol["species"], cfg["T"]), q296=sp["q296"])]))
    return out


def oracle_layer(cfg, mols, g):
    """cross sections of every line list (C oracle), absorption coefficient with the reference's
    expression (pyradClasses.py:583, 707-712) in NumPy"""
    from oracle import pyrad_oracle as orc
    jobs = []
    for m in mols:
        iso = m["isotopologues"][0]
        sel = orc.select_window(iso["lines"], g["eff_min"], g["eff_max"])
        jobs.append((sel, cfg["T"], cfg["P"], m["conc"], iso["molmass"], iso["q_T"], iso["q296"], g))
    res = oracle_xsecs(jobs)
    k = np.zeros(g["n_base"])
    for m, (xs, _, _) in zip(mols, res):
        k = k + orc.abs_coef(np.zeros(g["n_base"]) + xs, m["conc"], cfg["P"], cfg["T"])
    return res, k


@pytest.mark.parametrize("mode", ["exact", "budget"])
@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_whole_spectrum_cell_vs_c_oracle(ctx, workload, mode):
    """C2 (4e5 points, 65,536 lines) and C3 (2.4e6 points, 3 x 131,072 lines): every point of every cross
    section and of the absorption coefficient, default kernel (far-field series) and the all-direct kernel."""
    from oracle import pyrad_oracle as orc
    from pyrad_amd import engine
    cfg = synthetic.config_c2() if workload == "C2" else synthetic.config_c3()
    mols = mols_of(cfg)
    g = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], cfg["dynamic_resolution"])
    assert g["resolution"] == g["base_resolution"] and g["W"] == 5000
    ref, k_ref = oracle_layer(cfg, mols, g)
    xa = orc.x_axis(cfg["range_min"], cfg["range_max"], cfg["base_resolution"])
    tol = tolerance(mode, xa, cfg["T"], g["dfc"])
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                             cfg["base_resolution"], cfg["dynamic_resolution"])
    assert L.evals == sum(r[2] for r in ref)              # the metric's unit of work, counted by the oracle's own loop
    for variant in (5, 3):
        ctx.set_option("accum_variant", variant)
        ctx.set_option("accuracy", MODES[mode][0])
        try:
            L.enqueue(surface_T=288)
            r = L.results()
            for i, m in enumerate(mols):
                worst("%s %s variant %d %s xsec" % (workload, mode, variant, m["species"]), L.xsec_host(i), ref[i][0], tol)
            worst("%s %s variant %d abs_coef" % (workload, mode, variant), r["abs_coef"], k_ref, tol)
            tr = orc.transmittance(k_ref, cfg["depth"])
            # transmittance = exp(-k depth): an error of k of tol is an absolute error tol * k * depth of the exponent
            e = np.abs(r["transmittance"] - tr)
            assert np.all(e <= (tol * k_ref * cfg["depth"] + 4e-16) * tr + 1e-300)
            xs_planck = orc.transmission(tr, orc.planckWavenumber(xa, 288), orc.planckWavenumber(xa, cfg["T"]))
            e = np.abs(r["transmission"] - xs_planck)
            assert np.all(e <= ((tol * k_ref * cfg["depth"] + 4e-16) * tr + 2e-15) * np.maximum(orc.planckWavenumber(xa, 288), xs_planck))
            # the same cell through ONE merged accumulate job (lbl_layer_merged_step_dev): the absorption coefficient
            # sum_m f_m sum_iso xs_iso accumulated directly, no cross-section array written (the outputs are poisoned first)
            # (round 6: the far-field kernel's production shape exists in two builds - Gaussian runs of 16 points at four
            # waves per SIMD and of 32 points at three - and the library picks by the size of the launch: both are forced here)
            for grun in ((16, 32) if variant == 5 and workload == "C3" else (0,)):
                ctx.set_option("accum_gauss_run", grun)
                for b in (L.abs_coef, L.trans, L.I_out):
                    b.fill(float("nan"))
                L.enqueue(surface_T=288, merged=True)
                r = L.results()
                worst("%s %s variant %d MERGED (Gaussian runs: %d) abs_coef" % (workload, mode, variant, grun), r["abs_coef"], k_ref, tol)
                e = np.abs(r["transmittance"] - tr)
                assert np.all(e <= (tol * k_ref * cfg["depth"] + 4e-16) * tr + 1e-300)
                e = np.abs(r["transmission"] - xs_planck)
                assert np.all(e <= ((tol * k_ref * cfg["depth"] + 4e-16) * tr + 2e-15) * np.maximum(orc.planckWavenumber(xa, 288), xs_planck))
                if grun:
                    L.enqueue(surface_T=288)             # the per-list step through the same build
                    worst("%s %s variant %d per-list (Gaussian runs: %d) abs_coef" % (workload, mode, variant, grun),
                          L.results()["abs_coef"], k_ref, tol)
        finally:
            ctx.set_option("accum_variant", 5)
            ctx.set_option("accuracy", 0)
            ctx.set_option("accum_gauss_run", 0)
    L.free()


@pytest.mark.parametrize("step", ["per-list", "merged"])
@pytest.mark.parametrize("mode", ["exact", "budget"])
def test_whole_spectrum_column_all_layers_and_toa_vs_c_oracle(ctx, mode, step):
    """C5 at full size: the cross sections and the absorption coefficient of ALL 30 layers (windows from 5000 points
    down to 50: far-field kernel, the layers either side of the routing boundary at 640 points, skewed-range kernel)
    and the top-of-atmosphere radiance (the fold of pyradClasses.py:784-787 over the layers) at EVERY grid point.
    step "merged": one accumulate job per LAYER over its merged, factor-weighted line lists and the fold over the 30
    absorption coefficients (lbl_layers_merged_accumulate_dev + lbl_column_fold_dev); no cross section exists to compare.
    90 oracle jobs on a thread pool (ctypes releases the GIL).  The radiance is compared with an error bound
    propagated through the fold: an error tol k depth T of a layer's transmittance moves I by at most that times
    |I_in - B|, and T times what came in."""
    from oracle import pyrad_oracle as orc
    from oracle import c_oracle
    from pyrad_amd import engine
    c_oracle.load()
    col = synthetic.config_c5()
    cfgs = [dict(c, molecules=mols_of(c)) for c in col["layers"]]
    column = engine.ResidentColumn(ctx, cfgs, col["surface_T"])
    ctx.set_option("accuracy", MODES[mode][0])
    try:
        column.enqueue(layer_arrays=True, merged=(step == "merged"))
        ctx.sync()
    finally:
        ctx.set_option("accuracy", 0)
    grids = [orc.layer_grid(c["P"], c["range_min"], c["range_max"], c["base_resolution"], c["dynamic_resolution"]) for c in cfgs]
    jobs = []
    for c, g in zip(cfgs, grids):
        for m in c["molecules"]:
            iso = m["isotopologues"][0]
            sel = orc.select_window(iso["lines"], g["eff_min"], g["eff_max"])
            jobs.append((sel, c["T"], c["P"], m["conc"], iso["molmass"], iso["q_T"], iso["q296"], g))
    xa = orc.x_axis(cfgs[0]["range_min"], cfgs[0]["range_max"], cfgs[0]["base_resolution"])
    I_ref = orc.planckWavenumber(xa, col["surface_T"])
    I_bound = 4e-16 * I_ref * (2.0 + 1.4387773538277202 * xa / col["surface_T"])
    worst_all = 0.0
    with ThreadPoolExecutor(max_workers=14) as ex:
        futs = [ex.submit(c_oracle.create_cross_section_work, *j) for j in jobs]
        for li, (c, g) in enumerate(zip(cfgs, grids)):
            n_mol = len(c["molecules"])
            ref = [futs[li * n_mol + i].result() for i in range(n_mol)]
            tol = tolerance(mode, xa, c["T"], g["dfc"])
            Lr = column.layers[li]
            assert Lr.evals == sum(r[2] for r in ref)
            k_ref = np.zeros(g["n_base"])
            for i, m in enumerate(c["molecules"]):
                if step != "merged":
                    worst_all = max(worst_all, worst("C5 %s layer %d (W = %d) %s xsec" % (mode, li, g["W"], m["species"]),
                                                     Lr.jobs[i][3].download(column.n), ref[i][0], tol))
                k_ref = k_ref + orc.abs_coef(np.zeros(g["n_base"]) + ref[i][0], m["conc"], c["P"], c["T"])
            worst_all = max(worst_all, worst("C5 %s %s layer %d (W = %d) abs_coef" % (mode, step, li, g["W"]),
                                             Lr.abs_coef.download(column.n), k_ref, tol))
            tr = orc.transmittance(k_ref, c["depth"])
            B = orc.planckWavenumber(xa, c["T"])
            d_tr = (tol * k_ref * c["depth"] + 4e-16) * tr
            # the sweeps form the Planck exponent as n * (100 h c / k / T): two roundings placed differently from the
            # reference's operation order, worth b * 2^-52 relative on exp(b) (b = c2 n / T is up to 17 here)
            dB = 6e-16 * (1.0 + 1.4387773538277202 * xa / c["T"]) * B
            I_bound = tr * I_bound + d_tr * np.abs(I_ref - B) + 8e-16 * np.maximum(I_ref, B) + (1.0 - tr) * dB
            I_ref = orc.transmission(tr, I_ref, B)
            for i in range(n_mol):
                futs[li * n_mol + i] = None                    # (free the 19 MB arrays as we go)
    toa = column.results()["toa"]
    e = np.abs(toa - I_ref)
    i = int(np.argmax(e / I_bound))
    print("C5 " + mode + " " + step + " top-of-atmosphere radiance, every point: max rel err %.3e (point %d, bound there %.1e relative); "
          "worst cross section / absorption coefficient of the 30 layers %.3e"
          % (float(np.max(e / I_ref)), i, float(I_bound[i] / I_ref[i]), worst_all))
    assert np.all(e <= I_bound), (float((e / I_bound).max()), i)
    assert float(np.max(e / I_ref)) <= (1e-11 if mode == "exact" else 1e-9)
    column.free()


@pytest.mark.parametrize("W", [639, 640, 641, 642, 643, 738, 770])
def test_whole_spectrum_windows_at_the_kernel_routing_boundary(ctx, W):
    """Windows either side of the far-field kernel's lower limit (H = W - 2 < 640 points routes a line list to the
    skewed-range kernel) and the 738-point window of the column (spans that have no far line at all), through the
    DEFAULT routing on a grid large enough for it to engage (3 x 2.4e6 points): every point against the C oracle."""
    from oracle import pyrad_oracle as orc
    from pyrad_amd import engine
    P = (W - 0.5) * 0.001 * 1013.25 / 5.0
    cfg = synthetic.config_c3(n_lines=30000)
    cfg = dict(cfg, P=P, T=250)
    mols = mols_of(cfg)
    g = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], cfg["dynamic_resolution"])
    assert g["W"] == W and g["resolution"] == g["base_resolution"]
    ref, k_ref = oracle_layer(cfg, mols, g)
    xa = orc.x_axis(cfg["range_min"], cfg["range_max"], cfg["base_resolution"])
    tol = point_tolerance(xa, cfg["T"], g["dfc"])
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                             cfg["base_resolution"], cfg["dynamic_resolution"])
    assert L.evals == sum(r[2] for r in ref)
    L.enqueue(surface_T=288)
    lst, tabs, _ = ctx.schedule_export(0)
    skew = tabs.shape[0] * 512 >= 3 * g["n_work"] > tabs.shape[0] * 256          # spans of 512 points: the skewed-range kernel
    assert skew == (W - 2 < 640), (W, tabs.shape)
    for i, m in enumerate(mols):
        worst("W = %d (%s kernel) %s xsec" % (W, "skewed-range" if skew else "far-field", m["species"]), L.xsec_host(i), ref[i][0], tol)
    worst("W = %d abs_coef" % W, L.results()["abs_coef"], k_ref, tol)
    L.abs_coef.fill(float("nan"))
    L.enqueue(surface_T=288, merged=True)              # the same window through one merged accumulate job
    worst("W = %d MERGED abs_coef" % W, L.results()["abs_coef"], k_ref, tol)
    L.free()


@pytest.mark.parametrize("mode", ["exact", "budget"])
@pytest.mark.parametrize("G,rank", [(8, 4), (2, 1)])
def test_whole_spectrum_merged_shard_vs_c_oracle(ctx, G, rank, mode):
    """What rank `rank` of G computes in `bench.py --gpus G` (BASELINE config 4, merged step, cost-balanced bounds): its
    contiguous range of the 100-2500 cm^-1 cell from its halo of lines, ONE merged accumulate job, with the launch shape
    the library picks for a launch of that size (a shard of 8: four waves per span; the whole cell: unsplit spans with
    32-point Gaussian runs) - EVERY point of the range against the C oracle run on the same lines (round-5 verdict, item 3a)."""
    from oracle import pyrad_oracle as orc
    from pyrad_amd import engine
    cfg = synthetic.config_c3()
    mols = mols_of(cfg)
    g = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], cfg["dynamic_resolution"])
    plan = engine.balanced_shards([dict(cfg, molecules=mols)], G, rank)
    part = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                cfg["base_resolution"], cfg["dynamic_resolution"], shard=plan, keep_host_lines=True)
    sl = slice(part.first, part.first + part.count)
    jobs = [(lines, cfg["T"], cfg["P"], m["conc"], m["isotopologues"][0]["molmass"], m["isotopologues"][0]["q_T"],
             m["isotopologues"][0]["q296"], g) for m, lines in zip(mols, part._keep)]
    res = oracle_xsecs(jobs)
    k_ref = np.zeros(g["n_base"])
    for m, (xs, _, _) in zip(mols, res):
        k_ref = k_ref + orc.abs_coef(np.zeros(g["n_base"]) + xs, m["conc"], cfg["P"], cfg["T"])
    xa = orc.x_axis(cfg["range_min"], cfg["range_max"], cfg["base_resolution"])
    tol = tolerance(mode, xa, cfg["T"], g["dfc"])
    ctx.set_option("accuracy", MODES[mode][0])
    try:
        for b in (part.abs_coef, part.trans, part.I_out):
            b.fill(float("nan"))
        part.enqueue(surface_T=288, merged=True)
        r = part.results()
    finally:
        ctx.set_option("accuracy", 0)
    worst("C3 %s merged shard %d of %d (%d points) abs_coef" % (mode, rank, G, part.count), r["abs_coef"][sl], k_ref[sl], tol[sl])
    tr = orc.transmittance(k_ref, cfg["depth"])[sl]
    e = np.abs(r["transmittance"][sl] - tr)
    assert np.all(e <= (tol[sl] * k_ref[sl] * cfg["depth"] + 4e-16) * tr + 1e-300)
    part.free()
