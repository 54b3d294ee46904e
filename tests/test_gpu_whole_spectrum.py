"""GPU: EVERY grid point of the BASELINE configurations at full size against the plain-C oracle.

The sampled tests of test_gpu_fullsize.py look at a few dozen points; the accumulate kernel decides
its code path per span (edge / near / far / Gaussian-run classes, line split, worklist order), so a
mis-classified span can hide between samples.  oracle/lbl_oracle.c restates the reference's loop
(pyradClasses.py:361-407, pyradLineshape.py, pyradIntensity.py:16-32) at 2e8 evals/s per host core:
all of C2 in 3 s, each of C3's three line lists in ~6 s (one thread per list: ctypes releases the GIL,
no process is forked from the GPU process).  It is pinned to the goldens by tests/test_oracle_golden.py.

Tolerance: conftest.point_tolerance (per point: 2e-12 + the nu -> 0 amplification of the
stimulated-emission factor stated in ulps); the worst point of every comparison is printed.
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


@pytest.mark.parametrize("workload", ["C2", "C3"])
def test_whole_spectrum_cell_vs_c_oracle(ctx, workload):
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
    tol = point_tolerance(xa, cfg["T"], g["dfc"])
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                             cfg["base_resolution"], cfg["dynamic_resolution"])
    assert L.evals == sum(r[2] for r in ref)              # the metric's unit of work, counted by the oracle's own loop
    for variant in (5, 3):
        ctx.set_option("accum_variant", variant)
        try:
            L.enqueue(surface_T=288)
            r = L.results()
            for i, m in enumerate(mols):
                worst("%s variant %d %s xsec" % (workload, variant, m["species"]), L.xsec_host(i), ref[i][0], tol)
            worst("%s variant %d abs_coef" % (workload, variant), r["abs_coef"], k_ref, tol)
            tr = orc.transmittance(k_ref, cfg["depth"])
            # transmittance = exp(-k depth): an error of k of tol is an absolute error tol * k * depth of the exponent
            e = np.abs(r["transmittance"] - tr)
            assert np.all(e <= (tol * k_ref * cfg["depth"] + 4e-16) * tr + 1e-300)
        finally:
            ctx.set_option("accum_variant", 5)
    L.free()


def test_whole_spectrum_column_layers_vs_c_oracle(ctx):
    """C5 at full size: the cross sections and transmittance of layers 0 (W = 5000, far-field kernel), 14
    (W = 552, narrow-window kernel) and 29 (W = 49, the narrowest) at every grid point."""
    from oracle import pyrad_oracle as orc
    from pyrad_amd import engine
    col = synthetic.config_c5()
    cfgs = [dict(c, molecules=mols_of(c)) for c in col["layers"]]
    column = engine.ResidentColumn(ctx, cfgs, col["surface_T"])
    column.enqueue(layer_arrays=True)
    for li in (0, 14, 29):
        c = cfgs[li]
        g = orc.layer_grid(c["P"], c["range_min"], c["range_max"], c["base_resolution"], c["dynamic_resolution"])
        ref, k_ref = oracle_layer(c, c["molecules"], g)
        xa = orc.x_axis(c["range_min"], c["range_max"], c["base_resolution"])
        tol = point_tolerance(xa, c["T"], g["dfc"])
        Lr = column.layers[li]
        assert Lr.evals == sum(r[2] for r in ref)
        for i, m in enumerate(c["molecules"]):
            worst("C5 layer %d (W = %d) %s xsec" % (li, g["W"], m["species"]), Lr.jobs[i][3].download(column.n), ref[i][0], tol)
        worst("C5 layer %d abs_coef" % li, Lr.abs_coef.download(column.n), k_ref, tol)
    column.free()
