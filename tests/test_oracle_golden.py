"""Pin the CPU oracle (oracle/pyrad_oracle.py) against the golden vectors that
tests/golden/make_golden.py captured from the real reference."""
import numpy as np
import pytest

from conftest import load_golden, unpack_lines, rel_err
from oracle import pyrad_oracle as orc
from pyrad_amd import synthetic

TOL = 1e-12   # the restatement evaluates the reference's expressions in the reference's order


def run_cell(lines, species, conc, T, P, rmin, rmax, base, dyn, scalar=False):
    grid = orc.layer_grid(P, rmin, rmax, base, dyn)
    sel = orc.select_window(lines, grid["eff_min"], grid["eff_max"])
    fn = orc.create_cross_section_scalar if scalar else orc.create_cross_section
    sp = synthetic.SPECIES[species]
    xs, counts = fn(sel, T, P, conc, sp["molmass"], synthetic.q_value(species, T), sp["q296"], grid)
    return xs, counts, grid


def test_g0_known_answers():
    z = load_golden("G0_functions")
    assert float(z["c2"]) == orc.c2 == 1.4387773538277202
    assert orc.boltzmannFactors(1000, 250) == float(z["ka_boltzmann_1000_250"]) == 0.4088630124860266
    assert orc.stimulatedEmissions(667, 250) == float(z["ka_stimulated_667_250"]) == 1.018273023966143
    assert orc.intensityFactor(1e-20, 667, 250, 1000, 250, 286) == float(z["ka_intensity"]) == 4.7628629747218875e-21


def test_g0_intensity_functions_bit_exact():
    z = load_golden("G0_functions")
    for i, t in enumerate(z["T_list"]):
        assert np.array_equal(orc.boltzmannFactors(z["E"], t), z["boltzmann"][i])
        assert np.array_equal(orc.stimulatedEmissions(z["nu"], t), z["stimulated"][i])
        assert np.array_equal(orc.intensityFactor(z["S"], z["nu"], t, z["E"], z["q"][i], 286.09), z["intensity"][i])


def test_g0_halfwidths_and_shapes_bit_exact():
    z = load_golden("G0_functions")
    m = float(z["m"])
    for i, t in enumerate(z["T_list"]):
        assert np.array_equal(orc.gaussianHW(z["nu"], t, m), z["ghw"][i])
        assert np.array_equal(orc.lorentzHW(z["ga"], z["gs"], 1013.25, t, 4e-4, z["n_air"]), z["lhw"][i])
        assert np.array_equal(orc.lorentzHW(z["ga"], z["gs"], 10.0, t, .01, z["n_air"]), z["lhw_lowP"][i])
    for i, hw in enumerate(z["hw_l"]):
        assert np.array_equal(orc.lorentzLineShape(hw, z["x"]), z["lorentz"][i])
    for i, hw in enumerate(z["hw_g"]):
        assert np.array_equal(orc.gaussianLineShape(hw, z["x"]), z["gauss"][i])
        assert np.array_equal(orc.gaussianLineShape(hw, z["xf"]), z["gauss_fine"][i])
    for i, (g, l) in enumerate(zip(z["pv_g"], z["pv_l"])):
        assert np.array_equal(orc.pseudoVoigtShape(g, l, z["x"]), z["pvoigt"][i])
        assert np.array_equal(orc.pseudoVoigtShape(g, l, z["xf"]), z["pvoigt_fine"][i])


def test_g0_planck_bit_exact():
    z = load_golden("G0_functions")
    for i, t in enumerate((200, 288, 296, 320)):
        assert np.array_equal(orc.planckWavenumber(z["planck_n"], t), z["planck_wn"][i], equal_nan=True)
    assert np.isnan(z["planck_wn"][0][0])          # n = 0 -> 0/0 (pl:15 under pl:2)
    for i, t in enumerate((200.0, 288.0)):
        assert np.array_equal(orc.planckHz(z["planck_hz_x"], t), z["planck_hz"][i])
        assert np.array_equal(orc.planckWavelength(z["planck_lam_x"], t), z["planck_lam"][i])
    np.testing.assert_allclose(orc.planckWavenumber([600, 650, 700], 288),
                               [0.13515778, 0.13232193, 0.12759638], rtol=1e-7)


def test_g0_single_line_survey_known_answer():
    z = load_golden("G0_functions")
    lines = unpack_lines(z, "one.lines")
    xs, counts, grid = run_cell(lines, "co2", 400 * 10**-6, 296, 1013.25, 600, 700, .01, True, scalar=True)
    assert np.array_equal(xs, z["one.xsec"])
    assert float(z["one.lhw"][0]) == pytest.approx(0.070008, rel=1e-12)
    assert float(z["one.ghw"][0]) == 0.0007252622182303713
    assert float(z["one.broadened"][0]) == 650.0010000000001
    assert xs[5000] == 4.5462648814858876e-20
    nz = np.nonzero(xs)[0]
    assert (nz[0], nz[-1]) == (4502, 5498) and grid["W"] == 500
    assert counts == (0, 0, 1)


def test_g0_concentration_and_units():
    z = load_golden("G0_functions")
    got = [orc.concentration(ppm=400.0), orc.concentration(ppb=1.0), orc.concentration(ppb=1800.0),
           orc.concentration(**{"%": 1.0}), orc.concentration(concentration=0.0004)]
    assert np.array_equal(np.array(got), z["conc_values"])
    assert got[1] == 1e-08            # the ppb x 1e-8 quirk (cls:554)


@pytest.mark.parametrize("T", [296, 250])
def test_g1_cell(T):
    z = load_golden("G1_c1_cell")
    lines = unpack_lines(z, "lines")
    conc = orc.concentration(ppm=float(z["conc_ppm"]))
    xs, counts, grid = run_cell(lines, "co2", conc, T, float(z["P"]), 600, 700, .01, True)
    p = "T%d." % T
    assert grid["W"] == int(z["W"]) == 500
    lq = orc.line_quantities(orc.select_window(lines, grid["eff_min"], grid["eff_max"]), T, float(z["P"]), conc,
                             synthetic.SPECIES["co2"]["molmass"], 600, grid["resolution"])
    assert np.array_equal(lq["index"], z[p + "line_index"])
    # the reference evaluates these per line with Python scalars (libm pow); the
    # vectorised form may differ in the last bit
    assert rel_err(lq["lhw"], z[p + "line_lhw"]) <= 4e-16
    assert rel_err(lq["ghw"], z[p + "line_ghw"]) <= 4e-16
    assert rel_err(xs, z[p + "xsec"]) <= TOL
    kk = orc.abs_coef(xs, conc, float(z["P"]), T)
    assert rel_err(kk, z[p + "abs_coef"]) <= TOL
    tr = orc.transmittance(kk, float(z["depth"]))
    assert rel_err(tr, z[p + "transmittance"]) <= TOL
    xa = orc.x_axis(600, 700, .01)
    assert np.array_equal(xa, z["x_axis"])
    out = orc.transmission(tr, orc.planckWavenumber(xa, 288), orc.planckWavenumber(xa, T))
    assert rel_err(out, z[p + "transmission"]) <= TOL
    assert orc.integrateSpectrum(out, orc.pi, .01) == pytest.approx(float(z[p + "band_integral"]), rel=1e-13)
    if T == 296:
        assert rel_err(orc.absorbance(tr), z["absorbance"], floor=1e-300) <= 1e-9
        assert rel_err(orc.optical_depth(tr), z["optical_depth"], floor=1e-300) <= 1e-9
        assert np.max(np.abs((1 - tr) - z["emissivity"])) <= 1e-12            # cls:726-728


def test_g1_scalar_and_vector_forms_agree_bitwise():
    z = load_golden("G1_c1_cell")
    lines = {k: v[:300] for k, v in unpack_lines(z, "lines").items()}
    a, ca, _ = run_cell(lines, "co2", 4e-4, 296, 1013.25, 600, 700, .01, True, scalar=True)
    b, cb, _ = run_cell(lines, "co2", 4e-4, 296, 1013.25, 600, 700, .01, True, scalar=False)
    assert np.array_equal(a, b) and ca == cb


def test_g2_edges():
    z = load_golden("G2_edges")
    lines = unpack_lines(z, "lines")
    xs, _, grid = run_cell(lines, "co2", 4e-4, 296, 1013.25, 600, 700, .01, True, scalar=True)
    sel = orc.select_window(lines, grid["eff_min"], grid["eff_max"])
    lq = orc.line_quantities(sel, 296, 1013.25, 4e-4, synthetic.SPECIES["co2"]["molmass"], 600, .01)
    assert np.array_equal(lq["index"], z["line_index"])
    # truncation toward zero: nu in (min - res, min) lands on 0, not -1 (cls:390)
    idx = dict(zip(np.round(sel["nu"], 6), lq["index"]))
    assert idx[599.995] == 0 and idx[599.985] == -1 and idx[600.07] == 7 and idx[600.29] == 28
    assert rel_err(xs, z["xsec"]) <= TOL
    for i in range(len(lines["nu"])):
        one = {k: v[i:i + 1] for k, v in lines.items()}
        x1, _, _ = run_cell(one, "co2", 4e-4, 296, 1013.25, 600, 700, .01, True)
        assert rel_err(x1, z["single_xsec"][i]) <= TOL, i


def test_g3_pressure_ladder():
    z = load_golden("G3_pressure_ladder")
    expect = [(0.01, 500), (0.01, 247), (0.01, 50), (0.01, 5), (0.01, 1), (0.1, 500), (0.1, 987)]
    for j, P in enumerate(z["P_list"]):
        p = "P%d." % j
        lines = unpack_lines(z, p + "lines")
        xs, _, grid = run_cell(lines, "co2", 4e-4, 260, float(P), 640, 660, .01, True)
        assert (grid["resolution"], grid["W"]) == (float(z[p + "resolution"]), int(z[p + "W"]))
        assert (pytest.approx(grid["resolution"]), grid["W"]) == expect[j]
        assert grid["n_work"] == int(z[p + "n_work"])
        assert rel_err(xs, z[p + "xsec"]) <= TOL, j
        kk = orc.abs_coef(xs, 4e-4, float(P), 260)
        assert rel_err(kk, z[p + "abs_coef"]) <= TOL
        xa = orc.x_axis(640, 660, .01)
        out = orc.transmission(orc.transmittance(kk, 100.0), orc.planckWavenumber(xa, 288),
                               orc.planckWavenumber(xa, 260))
        assert rel_err(out, z[p + "transmission"]) <= TOL


def test_g4_regimes():
    z = load_golden("G4_regimes")
    a = unpack_lines(z, "a.lines")
    xs, counts, _ = run_cell(a, "co2", 4e-4, 296, 1013.25, 645, 655, .01, True)
    assert counts[1] > 20 and counts[2] > 20 and counts[0] == 0
    assert rel_err(xs, z["a.xsec"]) <= TOL
    b = unpack_lines(z, "b.lines")
    xs, counts, grid = run_cell(b, "co2", 4e-4, 220, 0.05, 650.0, 650.05, 1e-5, False)
    assert grid["W"] == int(z["b.W"]) and grid["W"] > 2
    assert counts[0] > 5 and counts[2] > 5          # Gaussian regime with live wings
    assert rel_err(xs, z["b.xsec"]) <= TOL
    c = unpack_lines(z, "c.lines")
    xs, counts, grid = run_cell(c, "co2", 4e-4, 220, 2.0, 650.0, 650.05, 1e-5, False)
    assert grid["W"] == int(z["c.W"]) and counts[2] == len(orc.select_window(c, grid["eff_min"], grid["eff_max"])["nu"])
    assert rel_err(xs, z["c.xsec"]) <= TOL


@pytest.mark.parametrize("tag,dyn", [("native", False), ("dynamic", True)])
def test_g5_native_and_regrid(tag, dyn):
    z = load_golden("G5_native_0p001")
    lines = unpack_lines(z, "lines")
    xs, _, grid = run_cell(lines, "co2", 4e-4, 296, 1013.25, 650, 660, .001, dyn)
    assert grid["W"] == int(z[tag + ".W"]) and grid["n_work"] == int(z[tag + ".n_work"])
    assert grid["resolution"] == float(z[tag + ".resolution"])
    assert np.array_equal(orc.x_axis(650, 660, .001), z[tag + ".x_axis"])
    assert rel_err(xs, z[tag + ".xsec"]) <= TOL
    assert rel_err(orc.abs_coef(xs, 4e-4, 1013.25, 296), z[tag + ".abs_coef"]) <= TOL


def test_g6_composition():
    z = load_golden("G6_composition")
    T, P, depth = int(z["T"]), float(z["P"]), float(z["depth"])
    conc = [orc.concentration(ppm=400), orc.concentration(**{"%": 1.5}), orc.concentration(ppb=1800)]
    assert np.array_equal(np.array(conc), z["concentration"])
    x0, _, _ = run_cell(unpack_lines(z, "co2.lines"), "co2", conc[0], T, P, 1000, 1040, .01, True)
    x1, _, _ = run_cell(unpack_lines(z, "co2_636.lines"), "co2_636", conc[0], T, P, 1000, 1040, .01, True)
    assert rel_err(x0, z["co2.iso0.xsec"]) <= TOL and rel_err(x1, z["co2.iso1.xsec"]) <= TOL
    assert rel_err(x0 + x1, z["co2.xsec"]) <= TOL           # no abundance weighting (cls:566-571)
    xh, _, _ = run_cell(unpack_lines(z, "h2o.lines"), "h2o", conc[1], T, P, 1000, 1040, .01, True)
    xc, _, _ = run_cell(unpack_lines(z, "ch4.lines"), "ch4", conc[2], T, P, 1000, 1040, .01, True)
    kk = np.zeros_like(x0)
    for xs, cc, name in ((x0 + x1, conc[0], "co2"), (xh, conc[1], "h2o"), (xc, conc[2], "ch4")):
        km = orc.abs_coef(xs, cc, P, T)
        assert rel_err(km, z[name + ".abs_coef"]) <= TOL
        kk = kk + km
    assert rel_err(kk, z["abs_coef"]) <= TOL
    tr = orc.transmittance(kk, depth)
    assert rel_err(tr, z["transmittance"]) <= TOL
    xa = orc.x_axis(1000, 1040, .01)
    out = orc.transmission(tr, orc.planckWavenumber(xa, 290), orc.planckWavenumber(xa, T))
    assert rel_err(out, z["transmission"]) <= TOL
    assert orc.integrateSpectrum(out, orc.pi, .01) == pytest.approx(float(z["band_integral"]), rel=1e-13)


def test_g7_column_fold():
    z = load_golden("G7_column")
    co2, h2o = unpack_lines(z, "co2.lines"), unpack_lines(z, "h2o.lines")
    xa = orc.x_axis(660, 680, .01)
    assert np.array_equal(orc.planckWavenumber(xa, 290), z["surface"])
    trs = []
    for i in range(3):
        T, P, depth = int(z["layer_T"][i]), float(z["layer_P"][i]), float(z["layer_depth"][i])
        cw = orc.concentration(percentage=float(z["h2o_perc"][i]))
        xc, _, _ = run_cell(co2, "co2", 4e-4, T, P, 660, 680, .01, True)
        xh, _, _ = run_cell(h2o, "h2o", cw, T, P, 660, 680, .01, True)
        kk = orc.abs_coef(xc, 4e-4, P, T) + orc.abs_coef(xh, cw, P, T)
        assert rel_err(kk, z["L%d.abs_coef" % i]) <= TOL
        trs.append(orc.transmittance(kk, depth))
        assert rel_err(trs[-1], z["L%d.transmittance" % i]) <= TOL
    out = orc.column_transmission(trs, [int(t) for t in z["layer_T"]], xa, 290)
    assert rel_err(out, z["L2.spectrum"]) <= TOL
    assert orc.integrateSpectrum(out, orc.pi, .01) == pytest.approx(float(z["toa_band_integral"]), rel=1e-13)


def test_eval_count_matches_loop():
    rng = np.random.default_rng(0)
    for W in (1, 2, 3, 7, 50):
        n = 40
        idx = rng.integers(-60, 100, size=30)
        brute = 0
        for c in idx.tolist():
            if 0 <= c <= n - 1:
                brute += 1
            for dx in range(1, W - 1):
                brute += (0 <= c + dx <= n - 1) + (0 <= c - dx <= n - 1)
        assert orc.eval_count(idx, W, n) == brute


def test_c_oracle_matches_numpy_oracle_and_golden():
    from oracle import c_oracle
    z = load_golden("G1_c1_cell")
    lines = unpack_lines(z, "lines")
    grid = orc.layer_grid(1013.25, 600, 700, .01, True)
    sel = orc.select_window(lines, grid["eff_min"], grid["eff_max"])
    sp = synthetic.SPECIES["co2"]
    for T in (296, 250):
        work, counts, evals = c_oracle.create_cross_section_work(sel, T, 1013.25, 4e-4, sp["molmass"],
                                                                 synthetic.q_value("co2", T), sp["q296"], grid)
        assert rel_err(work, z["T%d.xsec" % T]) <= TOL
        lq = orc.line_quantities(sel, T, 1013.25, 4e-4, sp["molmass"], 600, .01)
        assert evals == orc.eval_count(lq["index"], grid["W"], grid["n_work"])
        assert sum(counts) == len(sel["nu"])
    z5 = load_golden("G4_regimes")
    b = unpack_lines(z5, "b.lines")
    grid = orc.layer_grid(0.05, 650.0, 650.05, 1e-5, False)
    selb = orc.select_window(b, grid["eff_min"], grid["eff_max"])
    work, counts, _ = c_oracle.create_cross_section_work(selb, 220, 0.05, 4e-4, sp["molmass"],
                                                         synthetic.q_value("co2", 220), sp["q296"], grid)
    assert counts[0] > 5 and rel_err(work, z5["b.xsec"]) <= TOL


def test_g12_hitran_shaped_rows():
    """Rows of the kinds real HITRAN files hold beside the seeded lists' ranges (gamma_self = 0, gamma_air = 0 - both: the
    Gaussian-only branch over the full 500-point window at 1013 mbar, cls:379-381 - n_air < 0, delta_air > 0, E" = -1, S = 0,
    wavenumbers on exact grid multiples and on the window's ends, one wavenumber in two isotopologues, 1e-40 and 5e-16
    intensities, a 0.5 cm^-1 half-width), computed by the reference's own classes: NumPy and C restatements, per-line
    quantities, regime select, both isotopologues, two temperatures (round-5 verdict, item 3b)."""
    from oracle import c_oracle
    c_oracle.build(); c_oracle.load()
    z = load_golden("G12_hitran_shaped_rows")
    grid = orc.layer_grid(1013.25, 600, 700, .01, True)
    assert grid["W"] == 500
    for T in (296, 250):
        t = "T%d." % T
        k = np.zeros(grid["n_base"])
        xs_co2 = np.zeros(grid["n_base"])
        for tag, species, key in (("lines", "co2", "iso0"), ("lines2", "co2_636", "iso1")):
            sel = orc.select_window(unpack_lines(z, tag), grid["eff_min"], grid["eff_max"])
            sp = synthetic.SPECIES[species]
            xs, counts = orc.create_cross_section(sel, T, 1013.25, 4e-4, sp["molmass"], synthetic.q_value(species, T), sp["q296"], grid)[:2]
            assert rel_err(xs, z[t + key + ".xsec"]) <= TOL
            work, c_counts, _ = c_oracle.create_cross_section_work(sel, T, 1013.25, 4e-4, sp["molmass"], synthetic.q_value(species, T),
                                                                   sp["q296"], grid)
            assert rel_err(work, z[t + key + ".xsec"]) <= TOL and tuple(c_counts) == tuple(counts)
            if tag == "lines":
                assert tuple(counts) == tuple(np.bincount(z[t + "regime"], minlength=3)) and counts[0] == 3
                lq = orc.line_quantities(sel, T, 1013.25, 4e-4, sp["molmass"], 600, .01)
                # the reference's Line objects come out of a dict keyed by wavenumber: same order as the sorted selection
                assert np.array_equal(sel["nu"], z[t + "line_nu"]) and np.array_equal(lq["index"], z[t + "line_index"])
                assert rel_err(lq["lhw"], z[t + "line_lhw"]) <= 4e-16 and rel_err(lq["ghw"], z[t + "line_ghw"]) <= 4e-16
            xs_co2 = xs_co2 + xs
        k = k + orc.abs_coef(xs_co2, 4e-4, 1013.25, T)
        sp = synthetic.SPECIES["h2o"]
        sel = orc.select_window(unpack_lines(z, "h2o.lines"), grid["eff_min"], grid["eff_max"])
        xs = orc.create_cross_section(sel, T, 1013.25, 0.01, sp["molmass"], synthetic.q_value("h2o", T), sp["q296"], grid)[0]
        assert rel_err(xs, z[t + "h2o.xsec"]) <= TOL
        k = k + orc.abs_coef(xs, 0.01, 1013.25, T)
        assert rel_err(k, z[t + "abs_coef"]) <= TOL
    assert rel_err(xs_co2 * 0 + z["T296.co2.xsec"], z["T296.iso0.xsec"] + z["T296.iso1.xsec"]) <= 1e-15
    one = {f: v[unpack_lines(z, "lines")["nu"] == 612.34] for f, v in unpack_lines(z, "lines").items()}
    sp = synthetic.SPECIES["co2"]
    xs = orc.create_cross_section(one, 296, 1013.25, 4e-4, sp["molmass"], synthetic.q_value("co2", 296), sp["q296"], grid)[0]
    assert np.array_equal(xs != 0, z["gauss_only.xsec"] != 0) and rel_err(xs, z["gauss_only.xsec"]) <= TOL
    assert np.count_nonzero(xs) == 3              # 7e-4 cm^-1 wide on a 0.01 grid: the rest of the 500-point window underflows to 0
