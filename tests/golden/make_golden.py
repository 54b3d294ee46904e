#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (it needs /root/reference, which never travels
to the GPU box); its outputs, the ``*.npz`` files next to it, are committed and
are what the tests read.  The files hold data only: the inputs handed to the
reference and the numbers it returned.

Offline harness.  ``pyradUtilities`` cannot be imported as is (it
needs bs4 and fetches from hitran.org at import, ut:13, ut:1005) and
``pyradInteractive`` runs an input() loop at import (ui:761-762), so before
anything from /root/reference is imported:
  1. a stand-in module object named ``pyradUtilities`` is placed in sys.modules; it
     serves in-memory synthetic line lists / Q tables / molecule parameters in the
     reference's own schemas (pyrad_amd.synthetic) and exposes BASE_RESOLUTION;
  2. an empty stand-in ``pyradInteractive`` is placed in sys.modules;
  3. matplotlib uses the Agg backend;
  4. np.linspace is wrapped so that ``num=int(num)`` (the reference passes a float,
     cls:402-404, 704; NumPy >= 1.18 raises TypeError, older NumPy truncated).
The numeric modules pyradLineshape / pyradIntensity / pyradPlanck / pyradClasses
are then the reference's own files, executed unmodified.

Usage:  python tests/golden/make_golden.py        (rewrites every tests/golden/G*.npz)
"""
from __future__ import annotations

import io
import os
import sys
import types
import contextlib

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from pyrad_amd import synthetic  # noqa: E402

REFERENCE = "/root/reference"

# ----------------------------------------------------------------------------
# harness
# ----------------------------------------------------------------------------
_DATA = {}   # global_iso -> dict(lines=SoA, species=str)
_XSC_IDS = {}   # what the stand-in hands to cls:1024 (EXOTIC_IDS); filled by g8 from the xsc tree


def _install_harness():
    utils = types.ModuleType("pyradUtilities")
    utils.RES_MULTIPLIER = 1
    utils.BASE_RESOLUTION = .01 * utils.RES_MULTIPLIER
    utils.VERSION = "1.75"
    utils.returnXscTemperaturePressureValues = lambda: _XSC_IDS     # cls:1024 keeps this very dict
    utils.writeCurveToFile = lambda *a, **k: None

    def readMolParams(iso):
        return list(_DATA[iso]["params"])

    def gatherData(iso, lo, hi):
        return synthetic.lines_to_reference_dict(_DATA[iso]["lines"], lo, hi)

    def getQData(iso):
        return _DATA[iso]["q"]

    utils.readMolParams = readMolParams
    utils.gatherData = gatherData
    utils.getQData = getQData
    sys.modules["pyradUtilities"] = utils
    sys.modules["pyradInteractive"] = types.ModuleType("pyradInteractive")

    import matplotlib
    matplotlib.use("Agg")

    _linspace = np.linspace

    def linspace(start, stop, num=50, *a, **k):
        return _linspace(start, stop, int(num), *a, **k)

    np.linspace = linspace
    sys.path.insert(0, REFERENCE)
    with contextlib.redirect_stdout(io.StringIO()):
        import pyradLineshape, pyradIntensity, pyradPlanck, pyradClasses  # noqa: E401
    return utils, pyradLineshape, pyradIntensity, pyradPlanck, pyradClasses


UT, LS, INT, PL, CLS = _install_harness()


def _reset_reference_state():
    LS.cachedLorentz.clear(); LS.cachedGaussian.clear()
    LS.newLorentz.clear(); LS.newGaussian.clear()
    CLS.Layer.hasAtmosphere = False


def register(species, lines, q=None, params=None):
    sp = synthetic.SPECIES[species]
    _DATA[sp["global_iso"]] = dict(
        lines=lines, q=q if q is not None else synthetic.q_table(species),
        params=params if params is not None else synthetic.mol_params(species))


MOL_NAME = {"co2": "co2", "h2o": "h2o", "ch4": "ch4", "o3": "o3"}


def run_reference_layer(cfg, want=("xsec", "abs_coef", "transmittance")):
    """Build the reference's Layer from a config dict and pull results through the
    reference's own getters."""
    _reset_reference_state()
    UT.BASE_RESOLUTION = cfg["base_resolution"]
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        layer = CLS.Layer(cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"],
                          name=cfg.get("name", "golden"),
                          dynamicResolution=cfg.get("dynamic_resolution", True))
        for mol in cfg["molecules"]:
            species = mol["species"]
            register(species, mol["lines"])
            depth = mol.get("isotope_depth", 1)
            if depth == 2:
                register(species + "_636", mol["lines2"])
            layer.addMolecule(MOL_NAME[species], isotopeDepth=depth, **mol["conc"])
        out["abs_coef"] = np.array(CLS.getAbsCoef(layer))
        out["transmittance"] = np.array(CLS.getTransmittance(layer))
        out["xsec"] = [np.array(CLS.getCrossSection(m)) for m in layer]
        out["iso_xsec"] = [[np.array(CLS.getCrossSection(i)) for i in m] for m in layer]
        out["mol_abs_coef"] = [np.array(CLS.getAbsCoef(m)) for m in layer]
        out["x_axis"] = np.array(layer.xAxis)
        out["resolution"] = layer.resolution
        out["dfc"] = layer.distanceFromCenter
        out["W"] = len(np.arange(0, layer.distanceFromCenter, layer.resolution))
        out["n_work"] = len(layer.yAxis)
        out["concentration"] = [m.concentration for m in layer]
        if "surface_T" in cfg:
            surf = layer.planck(cfg["surface_T"])
            out["planck_surface"] = np.array(surf)
            out["planck_layer"] = np.array(layer.planck(layer.T))
            out["transmission"] = np.array(layer.transmission(surf))
            out["band_integral"] = float(CLS.integrateSpectrum(out["transmission"], CLS.pi,
                                                                 res=cfg["base_resolution"]))
            out["absorbance"] = np.array(CLS.getAbsorbance(layer))
            out["optical_depth"] = np.array(CLS.getOpticalDepth(layer))
            out["emissivity"] = np.array(CLS.getEmissivity(layer))
        # per-line derived quantities straight from the reference's Line properties
        first_iso = layer[0][0]
        out["line_nu"] = np.array([ln.wavenumber for ln in first_iso])
        out["line_lhw"] = np.array([ln.lorentzHW for ln in first_iso])
        out["line_ghw"] = np.array([ln.gaussianHW for ln in first_iso])
        out["line_broadened"] = np.array([ln.broadenedLine for ln in first_iso])
        out["line_index"] = np.array([int((ln.wavenumber - layer.rangeMin) / layer.resolution)
                                      for ln in first_iso], dtype=np.int64)
    UT.BASE_RESOLUTION = .01
    return out, layer


def pack_lines(prefix, lines):
    return {"%s.%s" % (prefix, f): lines[f] for f in synthetic.FIELDS}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def cfg_scalars(cfg):
    return dict(depth=np.float64(cfg["depth"]), T=np.int64(cfg["T"]), P=np.float64(cfg["P"]),
                range_min=np.float64(cfg["range_min"]), range_max=np.float64(cfg["range_max"]),
                base_resolution=np.float64(cfg["base_resolution"]),
                dynamic_resolution=np.bool_(cfg.get("dynamic_resolution", True)))


# ----------------------------------------------------------------------------
# G0: function-level vectors
# ----------------------------------------------------------------------------
def g0():
    rng = np.random.default_rng(100)
    T_list = np.array([200, 250, 296, 320], dtype=np.float64)
    E = rng.uniform(0, 5000, 64)
    nu = rng.uniform(50, 3000, 64)
    S = 10.0 ** rng.uniform(-28, -19, 64)
    out = dict(T_list=T_list, E=E, nu=nu, S=S, c2=np.float64(INT.c2))
    out["boltzmann"] = np.stack([INT.boltzmannFactors(E, t) for t in T_list])
    out["stimulated"] = np.stack([INT.stimulatedEmissions(nu, t) for t in T_list])
    q = 286.09 * (T_list / 296.0)
    out["q"] = q
    out["intensity"] = np.stack([INT.intensityFactor(S, nu, t, E, qq, 286.09) for t, qq in zip(T_list, q)])
    # known answers quoted in SURVEY.md §8c
    out["ka_boltzmann_1000_250"] = np.float64(INT.boltzmannFactors(1000, 250))
    out["ka_stimulated_667_250"] = np.float64(INT.stimulatedEmissions(667, 250))
    out["ka_intensity"] = np.float64(INT.intensityFactor(1e-20, 667, 250, 1000, 250, 286))
    # half-widths
    ga = rng.uniform(.05, .1, 64); gs = rng.uniform(.06, .12, 64); n = rng.uniform(.5, .8, 64)
    out.update(ga=ga, gs=gs, n_air=n)
    m = 43.98983 / 1000 / CLS.avo
    out["m"] = np.float64(m)
    out["ghw"] = np.stack([LS.gaussianHW(nu, t, m) for t in T_list])
    out["lhw"] = np.stack([LS.lorentzHW(ga, gs, 1013.25, t, 4e-4, n) for t in T_list])
    out["lhw_lowP"] = np.stack([LS.lorentzHW(ga, gs, 10.0, t, .01, n) for t in T_list])
    # line shapes on arange grids, a few half-widths per regime
    x = np.arange(0, 5.0, 0.01)
    xf = np.arange(0, 0.05, 0.001)
    hw_l = np.array([0.07, 0.0123, 0.5, 1e-3])
    hw_g = np.array([7.2e-4, 1e-3, 0.02, 0.3])
    out.update(x=x, xf=xf, hw_l=hw_l, hw_g=hw_g)
    LS.cachedLorentz.clear(); LS.cachedGaussian.clear()
    out["lorentz"] = np.stack([np.array(LS.lorentzLineShape(hw, x)) for hw in hw_l])
    out["gauss"] = np.stack([np.array(LS.gaussianLineShape(hw, x)) for hw in hw_g])
    out["gauss_fine"] = np.stack([np.array(LS.gaussianLineShape(hw, xf)) for hw in hw_g])
    pv_g = np.array([7.2e-4, 7.2e-4, 1e-3, 0.01, 0.05])
    pv_l = np.array([0.07, 7.2e-5, 1e-3, 0.02, 0.0006])
    out.update(pv_g=pv_g, pv_l=pv_l)
    out["pvoigt"] = np.stack([np.array(LS.pseudoVoigtShape(g, l, x)) for g, l in zip(pv_g, pv_l)])
    out["pvoigt_fine"] = np.stack([np.array(LS.pseudoVoigtShape(g, l, xf)) for g, l in zip(pv_g, pv_l)])
    LS.cachedLorentz.clear(); LS.cachedGaussian.clear(); LS.newLorentz.clear(); LS.newGaussian.clear()
    # planck
    pn = np.array([0.0, 1.0, 100.0, 600.0, 650.0, 700.0, 2500.0, 10000.0])
    out["planck_n"] = pn
    out["planck_wn"] = np.stack([PL.planckWavenumber(pn, t) for t in (200, 288, 296, 320)])
    hz = np.array([1e11, 1e12, 2e13, 1e14])
    lam = np.array([0.5, 4.0, 10.0, 15.0, 100.0])
    out.update(planck_hz_x=hz, planck_lam_x=lam)
    out["planck_hz"] = np.stack([PL.planckHz(hz, t) for t in (200.0, 288.0)])
    out["planck_lam"] = np.stack([PL.planckWavelength(lam, t) for t in (200.0, 288.0)])
    # the single CO2-like line of SURVEY §8c
    one = {k: np.array([v]) for k, v in dict(nu=650.003, sw=1e-20, a=1.0, elower=1000.0, gamma_air=.07,
                                             gamma_self=.09, delta_air=-.002, n_air=.7).items()}
    cfg = dict(depth=10.0, T=296, P=1013.25, range_min=600, range_max=700, base_resolution=.01,
               dynamic_resolution=True, molecules=[dict(species="co2", conc=dict(ppm=400), lines=one)])
    ref, _ = run_reference_layer(cfg)
    out.update({"one." + k: v for k, v in pack_lines("lines", one).items()})
    out["one.xsec"] = ref["xsec"][0]
    out["one.lhw"] = ref["line_lhw"]; out["one.ghw"] = ref["line_ghw"]
    out["one.broadened"] = ref["line_broadened"]
    # conversions / concentration setters (cls:121-156, 543-560)
    layer = CLS.Layer(1, 296, 1013.25, 600, 601, name="x")
    with contextlib.redirect_stdout(io.StringIO()):
        register("co2", one)
        mol = layer.addMolecule("co2", ppm=1)
        conc = []
        for kind, v in (("ppm", 400.0), ("ppb", 1.0), ("ppb", 1800.0), ("%", 1.0), ("concentration", 0.0004)):
            {"ppm": mol.setPPM, "ppb": mol.setPPB, "%": mol.setPercentage,
             "concentration": mol.setConcentration}[kind](v)
            conc.append(mol.concentration)
    out["conc_values"] = np.array(conc)
    out["convert"] = np.array([CLS.convertLength(2.0, "m"), CLS.convertLength(2.0, "ft"), CLS.convertLength(2.0, "in"),
                               CLS.convertPressure(2.0, "atm"), CLS.convertPressure(2.0, "bar"),
                               CLS.convertPressure(2.0, "pa"), CLS.convertRange(15.0, "um"),
                               CLS.convertTemperature(15.0, "C"), CLS.convertTemperature(59.0, "F")])
    save("G0_functions", **out)


# ----------------------------------------------------------------------------
# G1: C1-shaped gas cell at two temperatures
# ----------------------------------------------------------------------------
def g1():
    cfg = synthetic.config_c1(n_lines=2000)
    arrays = dict(cfg_scalars(cfg))
    arrays.update(pack_lines("lines", cfg["molecules"][0]["lines"]))
    arrays["conc_ppm"] = np.float64(400)
    for T in (296, 250):
        c = dict(cfg, T=T, surface_T=288)
        ref, _ = run_reference_layer(c)
        p = "T%d." % T
        arrays[p + "xsec"] = ref["xsec"][0]
        arrays[p + "abs_coef"] = ref["abs_coef"]
        arrays[p + "transmittance"] = ref["transmittance"]
        arrays[p + "transmission"] = ref["transmission"]
        arrays[p + "band_integral"] = np.float64(ref["band_integral"])
        arrays[p + "line_index"] = ref["line_index"]
        arrays[p + "line_lhw"] = ref["line_lhw"]
        arrays[p + "line_ghw"] = ref["line_ghw"]
        if T == 296:
            arrays["x_axis"] = ref["x_axis"]
            arrays["absorbance"] = ref["absorbance"]
            arrays["optical_depth"] = ref["optical_depth"]
            arrays["emissivity"] = ref["emissivity"]
            arrays["planck_surface"] = ref["planck_surface"]
            arrays["W"] = np.int64(ref["W"])
    save("G1_c1_cell", **arrays)


# ----------------------------------------------------------------------------
# G2: edge lines (index truncation, clipped wings, rounding-decided indices)
# ----------------------------------------------------------------------------
def g2():
    rmin, rmax = 600, 700
    nus = np.array([rmin - 4.5, rmin - 0.015, rmin - 0.005, 600.07, 600.29, 600.3, 612.345678,
                    650.0, 699.99, rmax - 0.001, 699.995, rmax + 0.004, rmax + 0.011, rmax + 4.9])
    n = len(nus)
    rng = np.random.default_rng(102)
    lines = dict(nu=nus, sw=10.0 ** rng.uniform(-22, -19, n), a=np.ones(n),
                 elower=rng.uniform(0, 3000, n), gamma_air=rng.uniform(.05, .1, n),
                 gamma_self=rng.uniform(.06, .12, n), delta_air=rng.uniform(-.01, 0, n),
                 n_air=rng.uniform(.5, .8, n))
    cfg = dict(depth=10.0, T=296, P=1013.25, range_min=rmin, range_max=rmax, base_resolution=.01,
               dynamic_resolution=True, molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines)])
    arrays = dict(cfg_scalars(cfg)); arrays.update(pack_lines("lines", lines))
    ref, _ = run_reference_layer(cfg)
    arrays["xsec"] = ref["xsec"][0]; arrays["abs_coef"] = ref["abs_coef"]
    arrays["line_index"] = ref["line_index"]
    # each edge line alone, so a wrong clip cannot hide under a neighbour
    singles = []
    for i in range(n):
        one = {k: v[i:i + 1] for k, v in lines.items()}
        r, _ = run_reference_layer(dict(cfg, molecules=[dict(species="co2", conc=dict(ppm=400), lines=one)]))
        singles.append(r["xsec"][0])
    arrays["single_xsec"] = np.stack(singles)
    save("G2_edges", **arrays)


# ----------------------------------------------------------------------------
# G3: pressure ladder (W from 1 to 987, including the regrid path)
# ----------------------------------------------------------------------------
def g3():
    arrays = {}
    Ps = np.array([1013.25, 500.0, 101.325, 10.1325, 1.0, 10132.5, 20000.0])
    arrays["P_list"] = Ps
    rmin, rmax = 640, 660
    for j, P in enumerate(Ps):
        lo, hi = synthetic.layer_window(P, rmin, rmax)
        lines = synthetic.make_lines(300 + j, 120, lo, hi)
        cfg = dict(depth=100.0, T=260, P=float(P), range_min=rmin, range_max=rmax, base_resolution=.01,
                   dynamic_resolution=True, surface_T=288,
                   molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines)])
        ref, _ = run_reference_layer(cfg)
        p = "P%d." % j
        arrays.update(pack_lines(p + "lines", lines))
        arrays[p + "xsec"] = ref["xsec"][0]
        arrays[p + "abs_coef"] = ref["abs_coef"]
        arrays[p + "transmission"] = ref["transmission"]
        arrays[p + "resolution"] = np.float64(ref["resolution"])
        arrays[p + "W"] = np.int64(ref["W"])
        arrays[p + "n_work"] = np.int64(ref["n_work"])
    arrays.update(depth=np.float64(100.0), T=np.int64(260), range_min=np.float64(rmin), range_max=np.float64(rmax),
                  base_resolution=np.float64(.01))
    save("G3_pressure_ladder", **arrays)


# ----------------------------------------------------------------------------
# G4: regimes (forced Lorentz, Voigt, Gaussian)
# ----------------------------------------------------------------------------
def g4():
    arrays = {}
    rmin, rmax = 645, 655
    # (a) 1 atm: gamma chosen so ratios straddle 100 -> Lorentz and Voigt both live
    lo, hi = synthetic.layer_window(1013.25, rmin, rmax)
    lines = synthetic.make_lines(400, 200, lo, hi)
    lines["gamma_air"] = np.linspace(0.04, 0.12, 200)
    cfg = dict(depth=10.0, T=296, P=1013.25, range_min=rmin, range_max=rmax, base_resolution=.01,
               dynamic_resolution=True, molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines)])
    ref, _ = run_reference_layer(cfg)
    arrays.update(pack_lines("a.lines", lines)); arrays["a.xsec"] = ref["xsec"][0]
    arrays["a.ratio"] = ref["line_lhw"] / ref["line_ghw"]
    # (b) fine base grid at low pressure: Gaussian and Voigt lines with W > 1
    #     BASE = 1e-5, P = 0.05 mbar -> dfc = 2.47e-4, W = 25; gamma spans both sides of ratio .01
    rmin2, rmax2 = 650.0, 650.05
    P = 0.05
    lo, hi = synthetic.layer_window(P, rmin2, rmax2)
    lines_b = synthetic.make_lines(401, 60, lo, hi, decimals=7)
    lines_b["gamma_air"] = np.linspace(0.02, 0.5, 60)
    cfg_b = dict(depth=1000.0, T=220, P=P, range_min=rmin2, range_max=rmax2, base_resolution=1e-5,
                 dynamic_resolution=False, molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines_b)])
    ref_b, _ = run_reference_layer(cfg_b)
    arrays.update(pack_lines("b.lines", lines_b)); arrays["b.xsec"] = ref_b["xsec"][0]
    arrays["b.ratio"] = ref_b["line_lhw"] / ref_b["line_ghw"]
    arrays["b.W"] = np.int64(ref_b["W"])
    arrays["b.abs_coef"] = ref_b["abs_coef"]
    # (c) the same grid at 2 mbar: Voigt with a live Gaussian core and W = 987
    P = 2.0
    lo, hi = synthetic.layer_window(P, rmin2, rmax2)
    lines_c = synthetic.make_lines(402, 40, lo, hi, decimals=7)
    cfg_c = dict(depth=1000.0, T=220, P=P, range_min=rmin2, range_max=rmax2, base_resolution=1e-5,
                 dynamic_resolution=False, molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines_c)])
    ref_c, _ = run_reference_layer(cfg_c)
    arrays.update(pack_lines("c.lines", lines_c)); arrays["c.xsec"] = ref_c["xsec"][0]
    arrays["c.ratio"] = ref_c["line_lhw"] / ref_c["line_ghw"]
    arrays["c.W"] = np.int64(ref_c["W"])
    save("G4_regimes", **arrays)


# ----------------------------------------------------------------------------
# G5: native 0.001 grid (W = 5000) and the dynamic-resolution interp path
# ----------------------------------------------------------------------------
def g5():
    arrays = {}
    rmin, rmax = 650, 660
    lo, hi = synthetic.layer_window(1013.25, rmin, rmax)
    lines = synthetic.make_lines(500, 150, lo, hi)
    arrays.update(pack_lines("lines", lines))
    for tag, dyn in (("native", False), ("dynamic", True)):
        cfg = dict(depth=10.0, T=296, P=1013.25, range_min=rmin, range_max=rmax, base_resolution=.001,
                   dynamic_resolution=dyn, surface_T=288,
                   molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines)])
        ref, _ = run_reference_layer(cfg)
        arrays[tag + ".xsec"] = ref["xsec"][0]
        arrays[tag + ".abs_coef"] = ref["abs_coef"]
        arrays[tag + ".transmission"] = ref["transmission"]
        arrays[tag + ".W"] = np.int64(ref["W"])
        arrays[tag + ".n_work"] = np.int64(ref["n_work"])
        arrays[tag + ".resolution"] = np.float64(ref["resolution"])
        arrays[tag + ".x_axis"] = ref["x_axis"]
    save("G5_native_0p001", **arrays)


# ----------------------------------------------------------------------------
# G6: composition (2 isotopologues, 3 molecules, % / ppm / ppb)
# ----------------------------------------------------------------------------
def g6():
    arrays = {}
    rmin, rmax = 1000, 1040
    lo, hi = synthetic.layer_window(800.0, rmin, rmax)
    l_co2 = synthetic.make_lines(600, 300, lo, hi)
    l_co2b = synthetic.make_lines(601, 150, lo, hi)
    l_h2o = synthetic.make_lines(602, 250, lo, hi)
    l_ch4 = synthetic.make_lines(603, 200, lo, hi)
    cfg = dict(depth=250.0, T=275, P=800.0, range_min=rmin, range_max=rmax, base_resolution=.01,
               dynamic_resolution=True, surface_T=290,
               molecules=[dict(species="co2", conc=dict(ppm=400), lines=l_co2, lines2=l_co2b, isotope_depth=2),
                          dict(species="h2o", conc={"%": 1.5}, lines=l_h2o),
                          dict(species="ch4", conc=dict(ppb=1800), lines=l_ch4)])
    ref, _ = run_reference_layer(cfg)
    arrays.update(cfg_scalars(cfg))
    arrays.update(pack_lines("co2.lines", l_co2)); arrays.update(pack_lines("co2_636.lines", l_co2b))
    arrays.update(pack_lines("h2o.lines", l_h2o)); arrays.update(pack_lines("ch4.lines", l_ch4))
    arrays["concentration"] = np.array(ref["concentration"])
    for i, name in enumerate(("co2", "h2o", "ch4")):
        arrays[name + ".xsec"] = ref["xsec"][i]
        arrays[name + ".abs_coef"] = ref["mol_abs_coef"][i]
    arrays["co2.iso0.xsec"] = ref["iso_xsec"][0][0]
    arrays["co2.iso1.xsec"] = ref["iso_xsec"][0][1]
    arrays["abs_coef"] = ref["abs_coef"]; arrays["transmittance"] = ref["transmittance"]
    arrays["transmission"] = ref["transmission"]; arrays["band_integral"] = np.float64(ref["band_integral"])
    save("G6_composition", **arrays)


# ----------------------------------------------------------------------------
# G7: 3-layer column fold of Layer.transmission (cls:784-787)
# ----------------------------------------------------------------------------
def g7():
    arrays = {}
    rmin, rmax = 660, 680
    specs = [(20000.0, 288, 1013.25), (50000.0, 262, 600.0), (120000.0, 231, 150.0)]   # depth cm, T, P
    lo, hi = synthetic.layer_window(1013.25, rmin, rmax)
    l_co2 = synthetic.make_lines(700, 260, lo, hi)
    l_h2o = synthetic.make_lines(701, 180, lo, hi)
    arrays.update(pack_lines("co2.lines", l_co2)); arrays.update(pack_lines("h2o.lines", l_h2o))
    arrays["layer_depth"] = np.array([s[0] for s in specs]); arrays["layer_T"] = np.array([s[1] for s in specs])
    arrays["layer_P"] = np.array([s[2] for s in specs])
    arrays["h2o_perc"] = np.array([1.0, 0.2, 0.01])
    arrays.update(range_min=np.float64(rmin), range_max=np.float64(rmax), base_resolution=np.float64(.01),
                  surface_T=np.int64(290))
    spectrum = None
    for i, (depth, T, P) in enumerate(specs):
        cfg = dict(depth=depth, T=T, P=P, range_min=rmin, range_max=rmax, base_resolution=.01,
                   dynamic_resolution=True,
                   molecules=[dict(species="co2", conc=dict(ppm=400), lines=l_co2),
                              dict(species="h2o", conc=dict(percentage=float(arrays["h2o_perc"][i])), lines=l_h2o)])
        ref, layer = run_reference_layer(cfg)
        if spectrum is None:
            spectrum = layer.planck(290)
            arrays["surface"] = np.array(spectrum)
        spectrum = layer.transmission(spectrum)
        arrays["L%d.transmittance" % i] = ref["transmittance"]
        arrays["L%d.abs_coef" % i] = ref["abs_coef"]
        arrays["L%d.spectrum" % i] = np.array(spectrum)
    arrays["toa_band_integral"] = np.float64(CLS.integrateSpectrum(spectrum, CLS.pi, res=.01))
    save("G7_column", **arrays)


# ----------------------------------------------------------------------------
# G8: measured cross-section ("xsc") molecules — ut:611-715, cls:165-233, cls:466-505
# ----------------------------------------------------------------------------
def _reference_xsc_functions(xsc_dir):
    """The reference's OWN xsc file functions, executed unmodified.  pyradUtilities cannot be
    imported (bs4, network at import), so the named top-level functions are cut out of its
    source text with ``ast`` and compiled into a namespace that supplies the module globals
    they read (xscDir, NULL_TAG, BASE_RESOLUTION, logToFile)."""
    import ast, re
    path = os.path.join(REFERENCE, "pyradUtilities.py")
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    wanted = {"openReturnLines", "parseXscFileName", "returnXscFileContents", "processXscFile",
              "returnXscTemperaturePressureValues", "writeXscFile"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {"os": os, "re": re, "np": np, "xscDir": xsc_dir, "NULL_TAG": '#/null/#',
          "BASE_RESOLUTION": .01, "logToFile": lambda *a, **k: None}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def g8():
    import json, shutil, tempfile
    arrays = {}
    root = tempfile.mkdtemp(prefix="pyrad_xsc_")
    try:
        ns = _reference_xsc_functions(root)
        # -- file names -> properties (ut:611-641) --------------------------------------------
        names = ["CFC11_296.0K-760.0Torr_620.0-680.0_0.01_air_12_34.txt",
                 "CFC12_250.0K-380.5Torr_590.0-710.0_0.05_N2_7_1.txt",
                 "HFC134a_273.1K-0.0Torr_600.0-700.0_0.01__00_3.txt",
                 "SF6_295K-700Torr_925-955_0.03_air_1_2.txt",
                 "CCl4_208.0K-7.5Torr_750.0-812.0_00.txt"]
        parsed = []
        for nm in names:
            try:
                parsed.append(ns["parseXscFileName"](nm))
            except Exception as e:                               # no trailing id -> False.replace
                parsed.append({"raises": type(e).__name__})
        arrays["names_json"] = np.array(json.dumps(names))
        arrays["parsed_json"] = np.array(json.dumps(parsed))

        # -- an xsc tree written by the reference's writer, read by its reader -----------------
        rng = np.random.default_rng(800)

        def table(lo, hi, res):
            x = np.arange(lo, hi, res)
            y = 1e-18 * np.exp(-((x - 0.5 * (lo + hi)) / (0.2 * (hi - lo))) ** 2) * (1 + 0.3 * rng.random(x.size))
            return x, y

        files = [("CFC11", 296.0, 760.0, 620.0, 680.0, .01, "air", "12-34"),
                 ("CFC12", 250.0, 380.5, 590.0, 710.0, .05, "N2", "7-1"),
                 ("CFC113", 296.0, 760.0, 590.0, 640.0, .01, "air", "2-2")]     # partial overlap with 600-700
        with contextlib.redirect_stdout(io.StringIO()):
            for mol, T, Ptorr, lo, hi, res, broad, ident in files:
                os.makedirs("%s/%s" % (root, mol), exist_ok=True)
                x, y = table(lo, hi, res)
                ns["BASE_RESOLUTION"] = res                       # writeXscFile puts it in the name (ut:538)
                ns["writeXscFile"](x, y, lo, hi, T, Ptorr, mol, "%s/%s" % (root, mol), broad, ident)
        ns["BASE_RESOLUTION"] = .01
        tree = {}
        for mol in sorted(os.listdir(root)):
            for fn in sorted(os.listdir("%s/%s" % (root, mol))):
                with open("%s/%s/%s" % (root, mol, fn), "rb") as f:
                    tree["%s/%s" % (mol, fn)] = f.read()
        arrays["tree_json"] = np.array(json.dumps(sorted(tree)))
        for i, key in enumerate(sorted(tree)):
            arrays["tree.%d" % i] = np.frombuffer(tree[key], dtype=np.uint8)
            mol, fn = key.split("/")
            got = ns["processXscFile"](mol, fn)
            arrays["read.%d.wavenumber" % i] = np.array(got["wavenumber"])
            arrays["read.%d.intensity" % i] = np.array(got["intensity"])
            arrays["read.%d.res" % i] = np.float64(got["res"])
        ids = ns["returnXscTemperaturePressureValues"]()
        arrays["exotic_ids_json"] = np.array(json.dumps(ids, sort_keys=True))

        # -- mergeArray / interpolateArray direct (cls:159-233) --------------------------------
        layer_axis = np.linspace(600, 700, 10000)
        cases = {
            "inside_new": (layer_axis, np.arange(620.0, 680.0, .01)),        # table inside the layer range
            "covers_new": (layer_axis, np.arange(590.0, 710.0, .01)),        # table wider than the layer
            "disjoint": (layer_axis, np.arange(720.0, 730.0, .01)),
            "left_partial": (layer_axis, np.arange(590.0, 640.0, .01)),      # starts before, ends inside
            "right_partial": (layer_axis, np.arange(660.0, 720.0, .01)),     # starts inside, ends after
            "skipped_value": (layer_axis, np.arange(650.0, 660.0, .01)),     # 650.0 is not on the rounded axis
            "short_table": (np.linspace(10, 11, 100), np.arange(10.5, 10.52, .01)),
            # the table overhangs a short axis on the left and ends with it: the zero padding count goes negative
            "overhang_a": (np.array([600.00, 600.01]), np.array([599.97, 599.98, 599.99, 600.00])),
            "overhang_b": (np.array([600.00, 600.01]), np.array([599.98, 599.99, 600.00, 600.01])),
            "overhang_c": (600.0 + 0.01 * np.arange(5), 599.95 + 0.01 * np.arange(9)),
        }
        with contextlib.redirect_stdout(io.StringIO()):
            for name, (nx, oxx) in cases.items():
                oy = 1.0 + rng.random(oxx.size)
                arrays["merge.%s.newX" % name] = nx
                arrays["merge.%s.oldX" % name] = oxx
                arrays["merge.%s.oldY" % name] = oy
                try:
                    arrays["merge.%s.out" % name] = np.asarray(CLS.mergeArray(nx, oxx, oy), dtype=np.float64)
                except Exception as e:
                    arrays["merge.%s.raises" % name] = np.array(type(e).__name__)
            arrays["interp.out"] = CLS.interpolateArray(np.arange(590.0, 710.0, .01), arrays["read.2.wavenumber"],
                                                        arrays["read.2.intensity"])
        arrays["merge_cases_json"] = np.array(json.dumps(list(cases)))

        # -- layers holding an xsc molecule, through the reference's own classes ---------------
        UT.processXscFile = ns["processXscFile"]
        UT.parseXscFileName = ns["parseXscFileName"]
        _XSC_IDS.clear(); _XSC_IDS.update(ids)
        lo, hi = synthetic.layer_window(1013.25, 600, 700)
        l_co2 = synthetic.make_lines(801, 300, max(lo - 0.01, 0), hi + 0.01)
        arrays.update(pack_lines("co2.lines", l_co2))
        layer_cases = [
            ("E1", dict(T=296, P=1013.25), ("CFC11", "CFC11_296.0K-760.0Torr_620.0-680.0_0.01_air_12_34.txt"), dict(ppb=25)),
            ("E2", dict(T=296, P=1013.25), ("CFC12", "CFC12_250.0K-380.5Torr_590.0-710.0_0.05_N2_7_1.txt"), dict(ppm=0.5)),
            ("E3", dict(T=280, P=900.0), ("CFC11", 0), dict(ppb=40)),            # file picked by index
            ("E4", dict(T=296, P=1013.25), ("CFC113", 0), dict(ppb=10)),         # wrong-length merge (cls:216-219)
        ]
        for tag, st, (mol, fn), conc in layer_cases:
            _reset_reference_state()
            UT.BASE_RESOLUTION = .01
            register("co2", l_co2)
            with contextlib.redirect_stdout(io.StringIO()):
                layer = CLS.Layer(10.0, st["T"], st["P"], 600, 700, name=tag)
                layer.addMolecule("co2", ppm=400)
                try:
                    m = layer.addMolecule({mol: fn}, **conc)
                except Exception as e:
                    arrays["%s.raises" % tag] = np.array(type(e).__name__)
                    continue
                arrays["%s.layer_T" % tag] = np.int64(layer.T)
                arrays["%s.layer_P" % tag] = np.float64(layer.P)
                arrays["%s.resolution" % tag] = np.float64(layer.resolution)
                arrays["%s.mol_xsec" % tag] = np.asarray(CLS.getCrossSection(m), dtype=np.float64)
                arrays["%s.mol_abs_coef" % tag] = np.asarray(CLS.getAbsCoef(m), dtype=np.float64)
                try:
                    arrays["%s.abs_coef" % tag] = np.asarray(CLS.getAbsCoef(layer), dtype=np.float64)
                except ValueError as e:           # a cross section of the wrong length cannot be summed
                    arrays["%s.abs_coef_raises" % tag] = np.array(type(e).__name__)
                    arrays["%s.spec_json" % tag] = np.array(json.dumps(dict(T=st["T"], P=st["P"], mol=mol, file=fn, conc=conc)))
                    continue
                arrays["%s.transmittance" % tag] = np.asarray(CLS.getTransmittance(layer), dtype=np.float64)
                arrays["%s.transmission" % tag] = np.asarray(layer.transmission(layer.planck(288)), dtype=np.float64)
                arrays["%s.layer_xsec" % tag] = np.asarray(CLS.getCrossSection(layer), dtype=np.float64)
            arrays["%s.spec_json" % tag] = np.array(json.dumps(dict(T=st["T"], P=st["P"], mol=mol, file=fn, conc=conc)))
        arrays["layer_cases_json"] = np.array(json.dumps([c[0] for c in layer_cases]))
    finally:
        shutil.rmtree(root, ignore_errors=True)
        _XSC_IDS.clear()
    save("G8_xsc", **arrays)


# ----------------------------------------------------------------------------
# G9: the data/ tree readers — ut:90-101 openReturnLines, ut:173-189 gatherData (segment loop),
#     ut:421-448 readHitranOnlineFile, ut:451-461 readQFile, ut:464-477 readMolParams
# ----------------------------------------------------------------------------
def _reference_data_dir_functions(root):
    """The reference's OWN cache-file readers, cut out of pyradUtilities.py with ``ast`` and executed
    unmodified (the module cannot be imported: bs4, network at import).  The namespace supplies the
    module globals they read; the download functions they would fall back to raise, so a test tree
    that made the reference reach for the network fails loudly instead."""
    import ast
    path = os.path.join(REFERENCE, "pyradUtilities.py")
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    wanted = {"openReturnLines", "readHitranOnlineFile", "readQFile", "readMolParams", "gatherData", "getQData"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted

    def no_network(*a, **k):
        raise AssertionError("the reference reached for a download: %r" % (a,))
    ns = {"os": os, "np": np, "cwd": root, "dataDir": root + "/data", "NULL_TAG": '#/null/#',
          "logToFile": lambda *a, **k: None, "downloadHitran": no_network, "downloadQData": no_network}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def _pyr_row(mol_id, local_iso, nu, sw, a, elower, g_air, g_self, d_air, n_air):
    # request_params order of ut:369-374: molec_id,local_iso_id,nu,sw,a,elower,gamma_air,gamma_self,delta_air,n_air
    return "%d,%d,%.6f,%.3E,%.3E,%.4f,%.4f,%.3f,%.6f,%.2f\n" % (mol_id, local_iso, nu, sw, a, elower, g_air, g_self, d_air, n_air)


def g9():
    import json, shutil, tempfile
    arrays = {}
    root = tempfile.mkdtemp(prefix="pyrad_data_")
    try:
        rng = np.random.default_rng(900)
        ns = _reference_data_dir_functions(root)
        files = {}

        def rows_for(lo, hi, n, mol_id, extra=()):
            nu = np.round(np.sort(rng.uniform(lo, hi, n)), 6)
            rows = [_pyr_row(mol_id, 1, v, 10.0 ** rng.uniform(-24, -19), rng.uniform(.01, 9), rng.uniform(0, 3000),
                             rng.uniform(.05, .1), rng.uniform(.06, .12), rng.uniform(-.01, 0), rng.uniform(.5, .8)) for v in nu]
            return rows + list(extra)

        def row_at(nu, sw, mol_id=2):
            return _pyr_row(mol_id, 1, nu, sw, 1.0, 100.0, .07, .09, -.002, .7)

        # isotopologue 7 (CO2 626): three 100 cm^-1 segments around a 600-700 layer at 1 atm (595-705)
        seg500 = rows_for(500.0, 600.0, 40, 2, [row_at(595.0, 1e-20), row_at(594.999999, 2e-20), row_at(595.000001, 3e-20),
                                                  row_at(600.0, 4e-20)])          # nu == window edge; 600.0 also in the next file
        seg600 = rows_for(600.0, 700.0, 160, 2, [row_at(600.0, 5e-20),              # duplicate nu ACROSS segments: last wins
                                                  row_at(650.123456, 6e-20), row_at(650.123456, 7e-20),   # ... and WITHIN one
                                                  row_at(700.0, 8e-20)])
        rng.shuffle(seg600)                                                         # unsorted rows
        seg700 = ["# a header line the cache writer might leave\n", "# and another\n"] + \
            rows_for(700.0, 800.0, 30, 2, [row_at(700.0, 9e-20), row_at(705.0, 1e-21), row_at(704.999999, 1.1e-21)])
        files["7/500.pyr"], files["7/600.pyr"], files["7/700.pyr"] = seg500, seg600, seg700
        # isotopologue 1 (H2O 161): a NULL_TAG segment (HITRAN had no lines there, ut:96-98), a regular one
        files["1/500.pyr"] = ["#/null/#\n"]
        files["1/600.pyr"] = rows_for(600.0, 700.0, 90, 1)
        files["1/700.pyr"] = rows_for(700.0, 800.0, 12, 1)
        for iso, species in ((7, "co2"), (1, "h2o")):
            q = synthetic.q_table(species, 400)
            files["%d/q%d.txt" % (iso, iso)] = ["%d %s\n" % (T, repr(float(Q))) for T, Q in sorted(q.items())]
            files["%d/params.pyr" % iso] = ["#\t#\t#\n", "# Molecule params for pyrad\n", "#\t#\t#\n",
                                            ",".join(str(x) for x in synthetic.mol_params(species)) + "\n"]
        tree = {}
        for rel, rows in files.items():
            full = os.path.join(root, "data", rel)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            with open(full, "w") as f:
                f.writelines(rows)
            with open(full, "rb") as f:
                tree[rel] = f.read()
        arrays["tree_json"] = np.array(json.dumps(sorted(tree)))
        for i, key in enumerate(sorted(tree)):
            arrays["tree.%d" % i] = np.frombuffer(tree[key], dtype=np.uint8)

        # -- the readers, query by query -------------------------------------------------------
        queries = [(7, 595.0, 705.0), (7, 650.5, 660.25), (7, 599.0, 601.0), (7, 0.0, 1000.0),
                   (1, 595.0, 705.0), (1, 510.0, 590.0), (1, 699.5, 700.5)]
        fields = ("isotope", "intensity", "einsteinA", "airHalfWidth", "selfHalfWidth", "lowerEnergy",
                  "tempExponent", "pressureShift")
        kept = []
        with contextlib.redirect_stdout(io.StringIO()):
            for qi, (iso, lo, hi) in enumerate(queries):
                if (iso, lo, hi) == (7, 0.0, 1000.0):
                    # segments 0..400 and 800, 900 do not exist: the reference would download them
                    try:
                        ns["gatherData"](iso, lo, hi)
                    except AssertionError:
                        arrays["q%d.raises" % qi] = np.array("download")
                    kept.append([iso, lo, hi])
                    continue
                got = ns["gatherData"](iso, lo, hi)
                arrays["q%d.nu" % qi] = np.array(list(got.keys()), dtype=np.float64)       # insertion order (cls:352 iterates it)
                for f in fields:
                    arrays["q%d.%s" % (qi, f)] = np.array([got[k][f] for k in got], dtype=np.float64)
                kept.append([iso, lo, hi])
            arrays["queries_json"] = np.array(json.dumps(kept))
            for iso in (7, 1):
                qd = ns["readQFile"](iso)
                arrays["qfile.%d.T" % iso] = np.array(list(qd.keys()), dtype=np.int64)
                arrays["qfile.%d.Q" % iso] = np.array(list(qd.values()), dtype=np.float64)
                arrays["params.%d_json" % iso] = np.array(json.dumps(ns["readMolParams"](iso)))

        # -- a layer fed by the reference's own readers, through the reference's own classes ---
        keep = (UT.gatherData, UT.getQData, UT.readMolParams)
        UT.gatherData, UT.getQData, UT.readMolParams = ns["gatherData"], ns["getQData"], ns["readMolParams"]
        try:
            for tag, T, P, rmin, rmax in (("L1", 296, 1013.25, 600, 700), ("L2", 250, 500.0, 640, 660)):
                _reset_reference_state()
                UT.BASE_RESOLUTION = .01
                with contextlib.redirect_stdout(io.StringIO()):
                    layer = CLS.Layer(25.0, T, P, rmin, rmax, name=tag)
                    layer.addMolecule("co2", ppm=400)
                    layer.addMolecule("h2o", percentage=1.2)
                    arrays["%s.abs_coef" % tag] = np.array(CLS.getAbsCoef(layer))
                    arrays["%s.transmittance" % tag] = np.array(CLS.getTransmittance(layer))
                    arrays["%s.transmission" % tag] = np.array(layer.transmission(layer.planck(288)))
                    arrays["%s.co2.xsec" % tag] = np.array(CLS.getCrossSection(layer[0]))
                    arrays["%s.h2o.xsec" % tag] = np.array(CLS.getCrossSection(layer[1]))
                    arrays["%s.n_lines" % tag] = np.array([len(layer[0][0]), len(layer[1][0])], dtype=np.int64)
                    arrays["%s.line_survey" % tag] = np.array(layer.lineSurvey)
                arrays["%s.spec_json" % tag] = np.array(json.dumps(dict(depth=25.0, T=T, P=P, rmin=rmin, rmax=rmax)))
            arrays["layer_cases_json"] = np.array(json.dumps(["L1", "L2"]))
        finally:
            UT.gatherData, UT.getQData, UT.readMolParams = keep
    finally:
        shutil.rmtree(root, ignore_errors=True)
    save("G9_data_dir", **arrays)


# ----------------------------------------------------------------------------
# G10: line survey — cls:409-428 createLineSurvey, cls:589-594 / 691-696 the molecule and layer sums
# ----------------------------------------------------------------------------
def g10():
    import json
    arrays = {}
    cases = []

    def survey_case(tag, cfg):
        _reset_reference_state()
        ref, layer = run_reference_layer(cfg)
        with contextlib.redirect_stdout(io.StringIO()):
            arrays["%s.layer" % tag] = np.array(layer.lineSurvey)
            for mi, m in enumerate(layer):
                arrays["%s.mol%d" % (tag, mi)] = np.array(m.lineSurvey)
                for ii, iso in enumerate(m):
                    arrays["%s.mol%d.iso%d" % (tag, mi, ii)] = np.array(iso.lineSurvey)
        arrays["%s.resolution" % tag] = np.float64(ref["resolution"])
        spec = dict(depth=cfg["depth"], T=cfg["T"], P=cfg["P"], range_min=cfg["range_min"], range_max=cfg["range_max"],
                    molecules=[dict(species=m["species"], conc=m["conc"], isotope_depth=m.get("isotope_depth", 1)) for m in cfg["molecules"]])
        arrays["%s.spec_json" % tag] = np.array(json.dumps(spec))
        for mi, m in enumerate(cfg["molecules"]):
            arrays.update(pack_lines("%s.mol%d.lines" % (tag, mi), m["lines"]))
            if m.get("isotope_depth", 1) == 2:
                arrays.update(pack_lines("%s.mol%d.lines2" % (tag, mi), m["lines2"]))
        cases.append(tag)

    # S1: the C1 cell (2000 lines on 10^4 bins: many bins hold several lines, summed in line order)
    survey_case("S1", synthetic.config_c1(n_lines=2000))
    # S2: edge lines of G2 (bins decided by truncation toward zero, lines outside the range dropped)
    rmin, rmax = 600, 700
    nus = np.array([rmin - 4.5, rmin - 0.015, rmin - 0.005, 600.07, 600.29, 600.3, 612.345678,
                    650.0, 699.99, rmax - 0.001, 699.995, rmax + 0.004, rmax + 0.011, rmax + 4.9])
    n = len(nus)
    rng = np.random.default_rng(102)
    lines = dict(nu=nus, sw=10.0 ** rng.uniform(-22, -19, n), a=np.ones(n), elower=rng.uniform(0, 3000, n),
                 gamma_air=rng.uniform(.05, .1, n), gamma_self=rng.uniform(.06, .12, n), delta_air=rng.uniform(-.01, 0, n),
                 n_air=rng.uniform(.5, .8, n))
    survey_case("S2", dict(depth=10.0, T=296, P=1013.25, range_min=rmin, range_max=rmax, base_resolution=.01,
                           dynamic_resolution=True, molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines)]))
    # S3: resolution != BASE (10132.5 mbar -> 0.1 cm^-1): the index uses the layer resolution while the array
    # has the base length (cls:416 vs 423), so the survey crowds into the first tenth of the array
    P = 10132.5
    lo, hi = synthetic.layer_window(P, 640, 660)
    survey_case("S3", dict(depth=100.0, T=260, P=P, range_min=640, range_max=660, base_resolution=.01,
                           dynamic_resolution=True,
                           molecules=[dict(species="co2", conc=dict(ppm=400), lines=synthetic.make_lines(305, 120, lo, hi))]))
    # S4: two isotopologues and three molecules (the sums of cls:589-594 and 691-696)
    lo, hi = synthetic.layer_window(800.0, 1000, 1040)
    survey_case("S4", dict(depth=50.0, T=280, P=800.0, range_min=1000, range_max=1040, base_resolution=.01,
                           dynamic_resolution=True,
                           molecules=[dict(species="co2", conc=dict(ppm=400), isotope_depth=2,
                                           lines=synthetic.make_lines(1061, 400, lo, hi), lines2=synthetic.make_lines(1062, 150, lo, hi)),
                                      dict(species="h2o", conc={"%": 1.5}, lines=synthetic.make_lines(1063, 300, lo, hi)),
                                      dict(species="ch4", conc=dict(ppb=1800), lines=synthetic.make_lines(1064, 200, lo, hi))]))
    arrays["cases_json"] = np.array(json.dumps(cases))
    save("G10_line_survey", **arrays)


# ----------------------------------------------------------------------------
# G11: the reference's own plot() and plotSpectrum() (cls:849-944) under the Agg backend: the
# curves they draw and the legend texts (band integrals rounded to 2 digits) read back from the figure
# ----------------------------------------------------------------------------
def g11():
    import json
    import matplotlib.pyplot as plt
    cfg = dict(synthetic.config_c1(n_lines=600), T=270, surface_T=288)

    def figure_curves():
        ax = plt.gcf().axes[0]
        lines = ax.get_lines()
        out = dict(labels=[ln.get_label() for ln in lines], x=[np.array(ln.get_xdata(), dtype=np.float64) for ln in lines],
                   y=[np.array(ln.get_ydata(), dtype=np.float64) for ln in lines],
                   xlabel=ax.get_xlabel(), ylabel=ax.get_ylabel(), title=ax.get_title(), yscale=ax.get_yscale())
        plt.close("all")
        return out

    _, layer = run_reference_layer(cfg)
    UT.BASE_RESOLUTION = cfg["base_resolution"]
    arrays = dict(cfg_scalars(cfg))
    arrays.update(pack_lines("lines", cfg["molecules"][0]["lines"]))
    arrays["conc_ppm"] = np.float64(400)
    meta = {}
    with contextlib.redirect_stdout(io.StringIO()):
        # createTransmission (ui:390-402): surface spectrum at the first temperature, the layer's own temperature appended
        temps = [288, layer.T]
        surface = PL.planckWavenumber(layer.xAxis, temps[0])
        CLS.plotSpectrum(layer, objList=[layer, layer[0]], surfaceSpectrum=surface, planckTemperatureList=temps)
        c = figure_curves()
        meta["transmission"] = {k: c[k] for k in ("labels", "xlabel", "ylabel", "title")}
        for i, y in enumerate(c["y"]):
            arrays["transmission.y%d" % i] = y
        arrays["transmission.x"] = c["x"][-1]
        arrays["surface"] = np.array(surface)
        # createPlanckCurves (ui:376-380): Planck curves only, the three abscissa types
        for kind, lo, hi in (("wavenumber", 600, 680), ("Hz", 1e12, 9e13), ("wavelength", 8, 40)):
            CLS.plotSpectrum(title="Planck spectrums", rangeMin=lo, rangeMax=hi, planckTemperatureList=[250, "300"], planckType=kind)
            c = figure_curves()
            meta["planck." + kind] = {k: c[k] for k in ("labels", "xlabel", "ylabel", "title")}
            meta["planck." + kind]["range"] = [lo, hi]
            for i, y in enumerate(c["y"]):
                arrays["planck.%s.y%d" % (kind, i)] = y
            arrays["planck.%s.x" % kind] = c["x"][0]
        # createPlot (ui:79-83): every plot type of the menu (ui:407-413) for the layer and its molecule
        for kind in ("transmittance", "absorption coefficient", "cross section", "absorbance", "optical depth", "line survey"):
            CLS.plot(kind, "golden %s" % kind, [layer, layer[0]])
            c = figure_curves()
            meta["plot." + kind] = {k: c[k] for k in ("labels", "xlabel", "ylabel", "title", "yscale")}
            for i, y in enumerate(c["y"]):
                if kind in ("transmittance", "line survey") or (kind == "optical depth" and i == 1):      # (the rest: G1 holds them)
                    arrays["plot.%s.y%d" % (kind, i)] = y
    UT.BASE_RESOLUTION = .01
    arrays["meta_json"] = np.array(json.dumps(meta))
    save("G11_plots", **arrays)



# ----------------------------------------------------------------------------
# G12: HITRAN-shaped rows (round-5 verdict, item 3b): what real .pyr rows hand to createCrossSection beside the seeded
# lists' gamma in [0.05, 0.12], n in [0.5, 0.8], delta <= 0 (ut:421-448 reads whatever the columns hold): gamma_self = 0,
# gamma_air = 0 (both: lorentzHW = 0 -> hwRatio < 0.01 -> the Gaussian-only branch over the full 500-point window at 1013
# mbar, cls:379-381), n_air < 0, delta_air > 0, E" = -1 (HITRAN's "unknown"), S = 0, wavenumbers on exact grid multiples and
# on the window's ends, the same wavenumber in two isotopologues, very weak and very strong lines, a very wide line.
# ----------------------------------------------------------------------------
def g12():
    rmin, rmax = 600, 700
    rng = np.random.default_rng(1200)
    special = [
        #  nu         sw       elower  g_air   g_self  d_air    n_air
        (612.340000, 3e-21,    500.0,  0.0,    0.0,    -0.002,  0.7),     # Gaussian-only, W = 500
        (612.345678, 2e-21,    800.0,  0.0,    0.0,     0.0,    0.0),     # Gaussian-only, no shift, n = 0
        (620.000000, 5e-22,   1200.0,  0.07,   0.0,    -0.003,  0.75),    # gamma_self = 0
        (620.010000, 4e-21,    300.0,  0.0,    0.09,   -0.001,  0.6),     # gamma_air = 0: lorentzHW = q * gamma_self (tiny)
        (630.500000, 1e-20,    900.0,  0.08,   0.1,    -0.004, -0.3),     # n_air < 0
        (631.250000, 8e-21,   1500.0,  0.06,   0.08,    0.006,  0.65),    # delta_air > 0
        (640.000000, 6e-21,     -1.0,  0.07,   0.09,   -0.002,  0.7),     # E" = -1
        (641.000000, 0.0,      700.0,  0.07,   0.09,   -0.002,  0.7),     # S = 0
        (650.000000, 2e-20,    100.0,  0.075,  0.095,  -0.001,  0.72),    # exact grid multiples
        (650.010000, 1e-20,    150.0,  0.065,  0.085,   0.0,    0.68),
        (650.020000, 3e-40,    150.0,  0.065,  0.085,  -0.002,  0.68),    # very weak
        (655.123456, 7e-21,   2000.0,  0.09,   0.11,   -0.005,  0.55),    # the same wavenumber in the second isotopologue
        (660.000000, 5e-16,     50.0,  0.05,   0.06,   -0.001,  0.5),     # very strong
        (670.300000, 2e-21,    400.0,  0.5,    0.6,    -0.002,  0.7),     # very wide
        (600.000000, 1e-20,    600.0,  0.07,   0.09,   -0.002,  0.7),     # on the window's lower end
        (699.990000, 1e-20,    600.0,  0.07,   0.09,   -0.002,  0.7),     # last grid point
        (700.000000, 1e-20,    600.0,  0.07,   0.09,   -0.002,  0.7),     # on the window's upper end (index N: outside)
        (596.000000, 4e-21,    600.0,  0.0,    0.0,    -0.002,  0.7),     # Gaussian-only line 4 cm^-1 below the window
    ]
    sp = np.array(special)
    n_fill = 60
    lo, hi = synthetic.layer_window(1013.25, rmin, rmax)
    fill = synthetic.make_lines(1201, n_fill, lo, hi)
    lines = dict(nu=np.concatenate([sp[:, 0], fill["nu"]]), sw=np.concatenate([sp[:, 1], fill["sw"]]),
                 a=np.ones(len(sp) + n_fill), elower=np.concatenate([sp[:, 2], fill["elower"]]),
                 gamma_air=np.concatenate([sp[:, 3], fill["gamma_air"]]), gamma_self=np.concatenate([sp[:, 4], fill["gamma_self"]]),
                 delta_air=np.concatenate([sp[:, 5], fill["delta_air"]]), n_air=np.concatenate([sp[:, 6], fill["n_air"]]))
    order = np.argsort(lines["nu"], kind="stable")
    lines = {k: v[order] for k, v in lines.items()}
    # second isotopologue: shares 655.123456 and 650.000000 with the first list, has rows of its own kind
    n2 = 40
    l2 = synthetic.make_lines(1202, n2, lo, hi)
    l2["nu"][5], l2["nu"][6] = 655.123456, 650.000000
    l2["gamma_self"][7] = 0.0
    l2["gamma_air"][8], l2["gamma_self"][8] = 0.0, 0.0
    l2["elower"][9] = -1.0
    l2["delta_air"][10] = 0.008
    l2["n_air"][11] = -0.5
    order = np.argsort(l2["nu"], kind="stable")
    l2 = {k: v[order] for k, v in l2.items()}
    arrays = {}
    for T in (296, 250):
        cfg = dict(depth=10.0, T=T, P=1013.25, range_min=rmin, range_max=rmax, base_resolution=.01,
                   dynamic_resolution=True, surface_T=288,
                   molecules=[dict(species="co2", conc=dict(ppm=400), lines=lines, lines2=l2, isotope_depth=2),
                              dict(species="h2o", conc={"%": 1.0}, lines=fill)])
        ref, layer = run_reference_layer(cfg)
        t = "T%d." % T
        arrays[t + "iso0.xsec"] = ref["iso_xsec"][0][0]
        arrays[t + "iso1.xsec"] = ref["iso_xsec"][0][1]
        arrays[t + "h2o.xsec"] = ref["xsec"][1]
        arrays[t + "abs_coef"] = ref["abs_coef"]
        if T == 296:                       # (the molecule's sum and the swept arrays once: the file stays below 1 MB)
            arrays[t + "co2.xsec"] = ref["xsec"][0]
            arrays[t + "transmittance"] = ref["transmittance"]
            arrays[t + "transmission"] = ref["transmission"]
        arrays[t + "line_lhw"] = ref["line_lhw"]; arrays[t + "line_ghw"] = ref["line_ghw"]
        arrays[t + "line_broadened"] = ref["line_broadened"]; arrays[t + "line_index"] = ref["line_index"]
        arrays[t + "line_nu"] = ref["line_nu"]
        # which branch the reference's regime select takes per line of the first list (cls:378-387)
        ratio = ref["line_lhw"] / ref["line_ghw"]
        arrays[t + "regime"] = np.where(ratio < 0.01, 0, np.where(ratio > 100, 1, 2)).astype(np.int64)
    arrays.update(cfg_scalars(cfg))
    arrays.update(pack_lines("lines", lines)); arrays.update(pack_lines("lines2", l2)); arrays.update(pack_lines("h2o.lines", fill))
    # the Gaussian-only line alone: its support is the full window (cls:379-381, 392-400)
    one = {k: v[lines["nu"] == 612.34] for k, v in lines.items()}
    r, _ = run_reference_layer(dict(cfg, T=296, molecules=[dict(species="co2", conc=dict(ppm=400), lines=one)]))
    arrays["gauss_only.xsec"] = r["xsec"][0]
    save("G12_hitran_shaped_rows", **arrays)


if __name__ == "__main__":
    if not os.path.isdir(REFERENCE):
        sys.exit("needs /root/reference (build container only)")
    which = sys.argv[1:] or ["g0", "g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12"]
    for name in which:
        globals()[name]()
