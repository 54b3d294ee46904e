"""GPU: the pyradClasses-shaped object model (pyrad_amd.model) against the golden vectors
captured from the reference's own Layer/Molecule objects."""
import numpy as np
import pytest

from conftest import load_golden, unpack_lines, rel_err
from pyrad_amd import synthetic

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture()
def pyrad():
    from pyrad_amd import model, data, settings, engine
    model.Layer.hasAtmosphere = False
    settings.set_resolution_multiplier(1)
    yield model
    settings.set_resolution_multiplier(1)
    data.set_source(None)


def source(**species_lines):
    from pyrad_amd import data
    return data.set_source(data.synthetic_source(species_lines))


def test_g1_layer_getters(pyrad):
    z = load_golden("G1_c1_cell")
    source(co2=unpack_lines(z, "lines"))
    layer = pyrad.Layer(float(z["depth"]), 296, 1013.25, 600, 700)
    mol = layer.addMolecule('co2', ppm=400)
    assert (layer.resolution, layer.distanceFromCenter) == (0.01, 5.0)
    assert len(mol[0]) == 2000 and mol.concText == '400 ppm'
    assert rel_err(pyrad.getCrossSection(mol), z["T296.xsec"]) <= RTOL
    assert rel_err(pyrad.getAbsCoef(layer), z["T296.abs_coef"]) <= RTOL
    assert rel_err(pyrad.getTransmittance(layer), z["T296.transmittance"]) <= RTOL
    assert rel_err(pyrad.getAbsorbance(layer), z["absorbance"], floor=1e-300) <= 1e-9
    assert rel_err(pyrad.getOpticalDepth(layer), z["optical_depth"], floor=1e-300) <= 1e-9
    assert np.max(np.abs(pyrad.getEmissivity(layer) - z["emissivity"])) <= 1e-11
    assert np.array_equal(layer.emittance, layer.emissivity)
    assert np.array_equal(layer.xAxis, z["x_axis"])
    surf = layer.planck(288)
    assert rel_err(surf, z["planck_surface"]) <= 1e-14
    spec = layer.transmission(surf)
    assert rel_err(spec, z["T296.transmission"]) <= RTOL
    assert pyrad.integrateSpectrum(spec, pyrad.pi) == pytest.approx(float(z["T296.band_integral"]), rel=1e-12)
    # dirty-flag protocol: temperature change -> recompute (cls:741-743), depth change -> nothing (cls:754)
    layer.changeTemperature(250)
    assert not mol.progressCrossSection and not mol[0].progressCrossSection
    assert rel_err(pyrad.getAbsCoef(layer), z["T250.abs_coef"]) <= RTOL
    assert mol.progressCrossSection
    layer.changeDepth(20.0)
    assert mol.progressCrossSection
    assert rel_err(pyrad.getTransmittance(layer), z["T250.transmittance"] ** 2) <= 1e-9
    with pytest.raises(KeyError):               # Q[T] needs an integer-K temperature (cls:389)
        layer.changeTemperature(296.5)
        pyrad.getAbsCoef(layer)


def test_g6_composition_objects(pyrad):
    z = load_golden("G6_composition")
    source(co2=unpack_lines(z, "co2.lines"), co2_636=unpack_lines(z, "co2_636.lines"),
           h2o=unpack_lines(z, "h2o.lines"), ch4=unpack_lines(z, "ch4.lines"))
    layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), 1000, 1040)
    co2 = layer.addMolecule('co2', isotopeDepth=2, ppm=400)
    h2o = layer.addMolecule('h2o', **{'%': 1.5})
    ch4 = layer.addMolecule(6, ppb=1800)
    assert ch4.name == 'CH4' and ch4.concentration == 1800 * 10**-8
    assert [m.concentration for m in layer] == list(z["concentration"])
    assert rel_err(pyrad.getAbsCoef(layer), z["abs_coef"]) <= RTOL
    assert rel_err(pyrad.getCrossSection(co2), z["co2.xsec"]) <= RTOL
    assert rel_err(pyrad.getCrossSection(co2[1]), z["co2.iso1.xsec"]) <= RTOL
    for m, name in ((co2, "co2"), (h2o, "h2o"), (ch4, "ch4")):
        assert rel_err(pyrad.getAbsCoef(m), z[name + ".abs_coef"]) <= RTOL
    assert rel_err(layer.transmission(layer.planck(290)), z["transmission"]) <= RTOL
    assert rel_err(pyrad.getCrossSection(layer), z["co2.xsec"] + z["h2o.xsec"] + z["ch4.xsec"]) <= 1e-13
    assert pyrad.returnPlot(layer, 'absorption coefficient')[1] == 0
    assert pyrad.returnPlot(layer, 'nonsense') is False
    copy = layer.returnCopy()
    assert isinstance(copy, pyrad.Layer) and [m.name for m in copy] == [m.name for m in layer]
    assert [m.concentration for m in copy] == [m.concentration for m in layer]
    assert all(m.layer is copy for m in copy)
    assert rel_err(pyrad.getAbsCoef(copy), z["abs_coef"]) <= RTOL


def test_g5_native_resolution_via_settings(pyrad):
    from pyrad_amd import settings
    z = load_golden("G5_native_0p001")
    source(co2=unpack_lines(z, "lines"))
    settings.set_resolution_multiplier(0.1)
    assert settings.BASE_RESOLUTION == pytest.approx(0.001)
    for tag, dyn in (("native", False), ("dynamic", True)):
        layer = pyrad.Layer(10.0, 296, 1013.25, 650, 660, dynamicResolution=dyn)
        layer.addMolecule('co2', ppm=400)
        assert layer.resolution == float(z[tag + ".resolution"])
        assert rel_err(pyrad.getAbsCoef(layer), z[tag + ".abs_coef"]) <= RTOL
        assert rel_err(layer.transmission(layer.planck(288)), z[tag + ".transmission"]) <= RTOL


def test_g3_change_pressure_reloads_window(pyrad):
    z = load_golden("G3_pressure_ladder")
    P = z["P_list"]
    source(co2=unpack_lines(z, "P0.lines"))
    layer = pyrad.Layer(100.0, 260, float(P[0]), 640, 660)
    layer.addMolecule('co2', ppm=400)
    assert rel_err(pyrad.getAbsCoef(layer), z["P0.abs_coef"]) <= RTOL
    source(co2=unpack_lines(z, "P5.lines"))
    fresh = pyrad.Layer(100.0, 260, float(P[5]), 640, 660)      # 10132.5 mbar -> resolution 0.1, regrid path
    fresh.addMolecule('co2', ppm=400)
    assert fresh.resolution == pytest.approx(0.1) and fresh.distanceFromCenter == pytest.approx(50.0)
    assert rel_err(pyrad.getAbsCoef(fresh), z["P5.abs_coef"]) <= RTOL
    # changePressure re-reads the lines (resetData, cls:752) but, like the reference, keeps the
    # effective range computed at construction (only changeRange updates it, cls:737-738)
    layer.changePressure(float(P[5]))
    assert layer.resolution == pytest.approx(0.1) and layer.distanceFromCenter == pytest.approx(50.0)
    assert (layer.effectiveRangeMin, layer.effectiveRangeMax) == (635.0, 665.0)
    assert not layer[0].progressCrossSection
    k_stale = pyrad.getAbsCoef(layer)
    assert k_stale.shape == (2000,) and np.all(k_stale <= z["P5.abs_coef"] * (1 + 1e-12))


def test_g7_atmosphere_column(pyrad):
    z = load_golden("G7_column")
    source(co2=unpack_lines(z, "co2.lines"), h2o=unpack_lines(z, "h2o.lines"))
    atm = pyrad.Atmosphere('column')
    for i in range(3):
        layer = atm.addLayer(float(z["layer_depth"][i]), int(z["layer_T"][i]), float(z["layer_P"][i]), 660, 680)
        layer.addMolecule('co2', ppm=400)
        layer.addMolecule('h2o', percentage=float(z["h2o_perc"][i]))
    assert atm.returnLayerNames() == ['Layer 1', 'Layer 2', 'Layer 3']
    toa = atm.transmission(surfaceTemperature=290)
    assert rel_err(toa, z["L2.spectrum"]) <= RTOL
    spec = atm[0].planck(290)
    for layer in atm:
        spec = layer.transmission(spec)
    assert rel_err(spec, z["L2.spectrum"]) <= RTOL


def test_g9_data_dir_to_layer(pyrad, tmp_path):
    """PyRad's on-disk inputs end to end: the data/ tree of G9 (duplicate wavenumbers within and across
    segment files, window-edge lines, '#' headers, a NULL_TAG segment, unsorted rows) read by
    PyradDataDir, through model.Layer and the HIP kernels, against what the reference's own classes
    computed on top of the reference's own readers from the same bytes."""
    import json
    from pyrad_amd import data
    from test_data_dir_golden_cpu import write_data_tree
    z = load_golden("G9_data_dir")
    write_data_tree(z, str(tmp_path))
    data.set_source(data.PyradDataDir(str(tmp_path)))
    for tag in json.loads(str(z["layer_cases_json"])):
        spec = json.loads(str(z["%s.spec_json" % tag]))
        pyrad.Layer.hasAtmosphere = False
        layer = pyrad.Layer(spec["depth"], spec["T"], spec["P"], spec["rmin"], spec["rmax"], name=tag)
        co2 = layer.addMolecule('co2', ppm=400)
        h2o = layer.addMolecule('h2o', percentage=1.2)
        assert [len(co2[0]), len(h2o[0])] == list(z["%s.n_lines" % tag])          # duplicates collapsed, edges excluded
        assert co2[0].q296 == 286.09 and h2o[0].molmass == 18.010565 and co2[0].q[296] == 286.09
        assert rel_err(pyrad.getAbsCoef(layer), z["%s.abs_coef" % tag]) <= RTOL
        assert rel_err(pyrad.getTransmittance(layer), z["%s.transmittance" % tag]) <= RTOL
        assert rel_err(layer.transmission(layer.planck(288)), z["%s.transmission" % tag]) <= RTOL
        assert rel_err(pyrad.getCrossSection(co2), z["%s.co2.xsec" % tag]) <= RTOL
        assert rel_err(pyrad.getCrossSection(h2o), z["%s.h2o.xsec" % tag]) <= RTOL
        # the survey adds S per bin in the READER's order; bins that hold several lines may differ in the last bit
        # from the reference's file-order sum when the file is unsorted
        assert rel_err(layer.lineSurvey, z["%s.line_survey" % tag]) <= 1e-15
        assert np.array_equal(layer.lineSurvey != 0, z["%s.line_survey" % tag] != 0)


def test_g10_line_survey_against_the_reference(pyrad):
    """K7 and the molecule / layer sums against the surveys the reference's createLineSurvey built
    (cls:409-428, 589-594, 691-696): several lines per bin, edge truncation, the resolution != BASE
    sizing quirk, two isotopologues and three molecules.  Bit for bit."""
    import json
    z = load_golden("G10_line_survey")
    for tag in json.loads(str(z["cases_json"])):
        spec = json.loads(str(z["%s.spec_json" % tag]))
        species_lines = {}
        for mi, m in enumerate(spec["molecules"]):
            species_lines[m["species"]] = unpack_lines(z, "%s.mol%d.lines" % (tag, mi))
            if m["isotope_depth"] == 2:
                species_lines[m["species"] + "_636"] = unpack_lines(z, "%s.mol%d.lines2" % (tag, mi))
        source(**species_lines)
        pyrad.Layer.hasAtmosphere = False
        layer = pyrad.Layer(spec["depth"], spec["T"], spec["P"], spec["range_min"], spec["range_max"], name=tag)
        for m in spec["molecules"]:
            layer.addMolecule(m["species"], isotopeDepth=m["isotope_depth"], **m["conc"])
        assert layer.resolution == float(z["%s.resolution" % tag])
        for mi, mol in enumerate(layer):
            for ii, iso in enumerate(mol):
                assert np.array_equal(iso.lineSurvey, z["%s.mol%d.iso%d" % (tag, mi, ii)]), (tag, mi, ii)
            assert np.array_equal(mol.lineSurvey, z["%s.mol%d" % (tag, mi)]), (tag, mi)
        assert np.array_equal(layer.lineSurvey, z["%s.layer" % tag]), tag
        assert np.array_equal(pyrad.returnPlot(layer, 'line survey')[0], z["%s.layer" % tag])


def test_line_survey_and_line_views(pyrad):
    z = load_golden("G2_edges")
    lines = unpack_lines(z, "lines")
    source(co2=lines)
    layer = pyrad.Layer(10.0, 296, 1013.25, 600, 700)
    mol = layer.addMolecule('co2', ppm=400)
    iso = mol[0]
    sel = np.sort(lines["nu"][(lines["nu"] > 595) & (lines["nu"] < 705)])
    assert [ln.wavenumber for ln in iso] == list(sel)
    survey = np.zeros(10000)
    order = np.argsort(lines["nu"], kind="stable")
    for i in order:
        if 595 < lines["nu"][i] < 705:
            idx = int((lines["nu"][i] - 600) / 0.01)
            if 0 <= idx <= 9999:
                survey[idx] = survey[idx] + lines["sw"][i]
    assert np.array_equal(layer.lineSurvey, survey)
    ln = iso[3]
    assert ln.lorentzHW > 0 and ln.gaussianHW > 0 and ln.broadenedLine < ln.wavenumber
    assert len(pyrad.totalLineList(layer)) == len(sel)


def test_no_source_and_xsc_are_loud(pyrad):
    from pyrad_amd import data
    data.set_source(None)
    layer = pyrad.Layer(10.0, 296, 1013.25, 600, 700)
    with pytest.raises(RuntimeError):
        layer.addMolecule('co2', ppm=400)
    source(co2=synthetic.make_lines(1, 10, 595, 705))
    with pytest.raises(RuntimeError):                      # no xsc source installed
        layer.addMolecule({'CFC-11': 'file.txt'}, ppb=1)


def test_g8_xsc_molecules(pyrad, tmp_path):
    """Layers holding a measured cross-section molecule next to CO2 lines (cls:466-505): the file
    sets the layer's T and P, its table is merged on the host, the sweep runs on the device."""
    import json
    from conftest import write_xsc_tree
    from pyrad_amd import data
    z = load_golden("G8_xsc")
    write_xsc_tree(z, str(tmp_path))
    source(co2=unpack_lines(z, "co2.lines"))
    data.set_xsc_source(data.XscDir(str(tmp_path)))
    try:
        for tag in json.loads(str(z["layer_cases_json"])):
            spec = json.loads(str(z["%s.spec_json" % tag]))
            pyrad.Layer.hasAtmosphere = False
            layer = pyrad.Layer(10.0, spec["T"], spec["P"], 600, 700, name=tag)
            layer.addMolecule('co2', ppm=400)
            m = layer.addMolecule({spec["mol"]: spec["file"]}, **spec["conc"])
            assert m.exotic and len(m) == 0 and m.name == spec["mol"]
            assert (layer.T, layer.P) == (int(z["%s.layer_T" % tag]), float(z["%s.layer_P" % tag]))
            assert layer.resolution == float(z["%s.resolution" % tag])
            assert np.array_equal(np.asarray(pyrad.getCrossSection(m), dtype=np.float64), z["%s.mol_xsec" % tag])
            if "%s.abs_coef_raises" % tag in z.files:          # a partial overlap gives a wrong-length table
                with pytest.raises(ValueError):
                    pyrad.getAbsCoef(layer)
                continue
            assert rel_err(pyrad.getAbsCoef(m), z["%s.mol_abs_coef" % tag]) <= RTOL
            assert rel_err(pyrad.getAbsCoef(layer), z["%s.abs_coef" % tag]) <= RTOL
            assert rel_err(pyrad.getTransmittance(layer), z["%s.transmittance" % tag]) <= RTOL
            assert rel_err(layer.transmission(layer.planck(288)), z["%s.transmission" % tag]) <= RTOL
            assert rel_err(pyrad.getCrossSection(layer), z["%s.layer_xsec" % tag]) <= RTOL
            # the measured table survives what invalidates line-by-line cross sections (cls:40)
            before = np.array(m.crossSection)
            layer.changeTemperature(layer.T)
            m.setPPB(3)
            assert m.progressCrossSection and np.array_equal(m.crossSection, before)
            copy = m.returnCopy()
            assert copy.exotic and np.array_equal(copy.crossSection, before)
    finally:
        data.set_xsc_source(None)


@pytest.mark.parametrize("seed", range(4))
def test_lazy_protocol_random_sequences(pyrad, seed):
    """The dirty-flag protocol (cls:32-88, 543-560, 734-755) under random sequences of the mutators the
    interactive menu offers: after every operation the layer's absorption coefficient and transmittance
    equal those of a layer built from scratch in the current state."""
    rng = np.random.default_rng(4000 + seed)
    lines = dict(co2=synthetic.make_lines(41, 500, 580, 720), h2o=synthetic.make_lines(42, 300, 580, 720))
    source(**lines)

    def fresh(state):
        pyrad.Layer.hasAtmosphere = False
        L = pyrad.Layer(state["depth"], state["T"], state["P"], state["rmin"], state["rmax"])
        L.addMolecule('co2', ppm=state["co2"])
        L.addMolecule('h2o', percentage=state["h2o"])
        return pyrad.getAbsCoef(L), pyrad.getTransmittance(L)

    state = dict(depth=10.0, T=296, P=1013.25, rmin=600, rmax=700, co2=400.0, h2o=1.0)
    layer = pyrad.Layer(state["depth"], state["T"], state["P"], state["rmin"], state["rmax"])
    co2 = layer.addMolecule('co2', ppm=state["co2"])
    h2o = layer.addMolecule('h2o', percentage=state["h2o"])
    for step in range(12):
        op = int(rng.integers(0, 6))
        if op == 0:
            state["T"] = int(rng.integers(200, 330)); layer.changeTemperature(state["T"])
        elif op == 1:
            state["depth"] = float(rng.uniform(1.0, 5000.0)); layer.changeDepth(state["depth"])
        elif op == 2:
            state["co2"] = float(rng.uniform(100.0, 900.0)); co2.setPPM(state["co2"])
        elif op == 3:
            state["h2o"] = float(rng.uniform(0.01, 3.0)); h2o.setPercentage(state["h2o"])
        elif op == 4:
            state["P"] = float(rng.choice([1013.25, 700.0, 300.0, 120.0])); layer.changePressure(state["P"])
        else:
            state["rmin"] = int(rng.choice([600, 620])); state["rmax"] = state["rmin"] + int(rng.choice([60, 80]))
            layer.changeRange(state["rmin"], state["rmax"])
        k, t = pyrad.getAbsCoef(layer), pyrad.getTransmittance(layer)
        kf, tf = fresh(state)
        assert np.array_equal(k, kf) and np.array_equal(t, tf), (seed, step, op, state)


def test_molecule_sum_is_refreshed_after_an_isotopologue_changes(pyrad):
    """Molecule.createCrossSection re-sums its isotopologues on every call (cls:566-571): after an isotopologue
    has been recomputed, or has had a host array assigned, the molecule's (and then the layer's) cross section
    must be the sum of the CURRENT isotopologue cross sections, not a cached earlier one."""
    z = load_golden("G6_composition")
    source(co2=unpack_lines(z, "co2.lines"), co2_636=unpack_lines(z, "co2_636.lines"))
    layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), 1000, 1040)
    co2 = layer.addMolecule('co2', isotopeDepth=2, ppm=400)
    first = np.array(pyrad.getCrossSection(co2))
    assert rel_err(first, z["co2.xsec"]) <= RTOL
    # recompute one isotopologue, then the molecule again: same physics, the sum must still be right
    co2[1].createCrossSection()
    co2.createCrossSection()
    assert np.array_equal(np.array(co2.crossSection), first)
    # a host array assigned to an isotopologue: the next molecule sum carries it
    doubled = 2.0 * np.array(co2[1].crossSection)
    co2[1].crossSection = doubled
    co2.createCrossSection()
    want = np.zeros_like(first) + np.array(co2[0].crossSection) + doubled
    assert np.array_equal(np.array(co2.crossSection), want)
    # ... and so does the layer's cross section, through the molecule's
    layer.createCrossSection()
    assert np.array_equal(np.array(layer.crossSection), np.zeros_like(first) + want)
    # a user-assigned molecule cross section survives getters that do not recreate it
    co2.crossSection = first
    pyrad.getAbsCoef(layer)
    assert co2.crossSection is first


def test_g11_plot_and_plot_spectrum_like_the_reference(pyrad):
    """pyrad.plot (cls:849-873) and pyrad.plotSpectrum (cls:876-944) as pyradInteractive calls them (ui:83, 373, 399):
    curves, axis labels, titles and legend texts (band integrals rounded to two digits) against what the
    reference's own functions drew under the Agg backend (G11)."""
    import json
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    z = load_golden("G11_plots")
    meta = json.loads(str(z["meta_json"]))
    source(co2=unpack_lines(z, "lines"))
    layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), float(z["range_min"]), float(z["range_max"]), name="C1")
    co2 = layer.addMolecule('co2', ppm=int(z["conc_ppm"]))

    def drawn():
        ax = plt.gcf().axes[0]
        out = dict(labels=[ln.get_label() for ln in ax.get_lines()], y=[np.asarray(ln.get_ydata()) for ln in ax.get_lines()],
                   x=[np.asarray(ln.get_xdata()) for ln in ax.get_lines()],
                   xlabel=ax.get_xlabel(), ylabel=ax.get_ylabel(), title=ax.get_title(), yscale=ax.get_yscale())
        plt.close("all")
        return out

    # createTransmission (ui:390-402)
    temps = [288, layer.T]
    surface = layer.planck(temps[0])
    assert rel_err(surface, z["surface"]) <= RTOL
    pyrad.plotSpectrum(layer, objList=[layer, co2], surfaceSpectrum=surface, planckTemperatureList=temps)
    got, want = drawn(), meta["transmission"]
    assert (got["labels"], got["xlabel"], got["ylabel"], got["title"]) == (want["labels"], want["xlabel"], want["ylabel"], want["title"])
    for i in range(4):
        assert rel_err(got["y"][i], z["transmission.y%d" % i]) <= RTOL, i
    assert np.array_equal(got["x"][-1], z["transmission.x"])
    # the same numbers without a figure
    spec = pyrad.spectrumCurves(layer, objList=[layer, co2], surfaceSpectrum=surface, planckTemperatureList=temps)
    assert [c["label"] for c in spec["curves"]] == want["labels"] and spec["surfacePower"] > 0
    # createPlanckCurves (ui:376-380)
    for kind in ("wavenumber", "Hz", "wavelength"):
        want = meta["planck." + kind]
        lo, hi = want["range"]
        pyrad.plotSpectrum(title="Planck spectrums", rangeMin=lo, rangeMax=hi, planckTemperatureList=[250, "300"], planckType=kind)
        got = drawn()
        assert (got["labels"], got["xlabel"], got["ylabel"], got["title"]) == (want["labels"], want["xlabel"], want["ylabel"], want["title"]), kind
        for i in range(2):
            assert rel_err(got["y"][i], z["planck.%s.y%d" % (kind, i)]) <= RTOL, (kind, i)
        assert np.array_equal(got["x"][0], z["planck.%s.x" % kind])
    # createPlot (ui:79-83): every plot type of the menu
    for kind in ("transmittance", "absorption coefficient", "cross section", "absorbance", "optical depth", "line survey"):
        want = meta["plot." + kind]
        pyrad.plot(kind, "golden %s" % kind, [layer, co2])
        got = drawn()
        assert all(got[k] == want[k] for k in ("labels", "xlabel", "ylabel", "title", "yscale")), kind
        for i in range(2):
            key = "plot.%s.y%d" % (kind, i)
            if key in z.files:
                assert rel_err(got["y"][i], z[key]) <= (0.0 if kind == "line survey" else RTOL), key


def test_merged_layer_step_is_lazy_about_cross_sections(pyrad):
    """settings.LAYER_STEP "merged" (the default): getAbsCoef(layer) runs ONE accumulate job over the layer's merged,
    factor-weighted line lists and writes no isotopologue cross section; the reference's protocol flags
    (progressCrossSection, cls:32-88) read as if it had, and getCrossSection(isotope | molecule | layer) produces the
    arrays on demand through the per-line-list path - bit for bit what a per-list layer computes - in either order,
    also after mutators."""
    from pyrad_amd import settings
    z = load_golden("G6_composition")
    src = dict(co2=unpack_lines(z, "co2.lines"), co2_636=unpack_lines(z, "co2_636.lines"),
               h2o=unpack_lines(z, "h2o.lines"), ch4=unpack_lines(z, "ch4.lines"))
    source(**src)

    def build():
        pyrad.Layer.hasAtmosphere = False
        layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), 1000, 1040)
        layer.addMolecule('co2', isotopeDepth=2, ppm=400)
        layer.addMolecule('h2o', **{'%': 1.5})
        layer.addMolecule(6, ppb=1800)
        return layer

    assert settings.LAYER_STEP == "merged"
    settings.set_layer_step("per-list")
    try:
        ref = build()
        k_ref = np.array(pyrad.getAbsCoef(ref))
        xs_ref = [[np.array(pyrad.getCrossSection(iso)) for iso in m] for m in ref]
        mol_ref = [np.array(pyrad.getCrossSection(m)) for m in ref]
        ref.changeTemperature(250)
        k_ref250 = np.array(pyrad.getAbsCoef(ref))
        xs_ref250 = np.array(pyrad.getCrossSection(ref[0][1]))
    finally:
        settings.set_layer_step("merged")
    assert rel_err(k_ref, z["abs_coef"]) <= RTOL

    # layer first, cross sections afterwards
    layer = build()
    k = np.array(pyrad.getAbsCoef(layer))
    assert rel_err(k, z["abs_coef"]) <= RTOL and rel_err(k, k_ref) <= 1e-13
    assert all(m.progressCrossSection for m in layer)                       # as after the reference's Layer.absCoef
    assert all(iso.progressCrossSection and iso._xs_deferred and not iso._dev_xsec_valid for m in layer for iso in m)
    assert np.array_equal(pyrad.getCrossSection(layer[0][1]), xs_ref[0][1])      # made now, by the per-list path
    assert not layer[0][1]._xs_deferred and layer[0][0]._xs_deferred
    assert np.array_equal(pyrad.getAbsCoef(layer), k)                       # nothing changed: the merged arrays stand
    for m, want, isos in zip(layer, mol_ref, xs_ref):
        assert np.array_equal(pyrad.getCrossSection(m), want)
        for iso, w in zip(m, isos):
            assert np.array_equal(iso.crossSection, w)
    assert rel_err(pyrad.getCrossSection(layer), z["co2.xsec"] + z["h2o.xsec"] + z["ch4.xsec"]) <= 1e-13
    # every cross section is current now: the layer's chain may come from the sweep over them (per-list arithmetic)
    assert np.array_equal(pyrad.getAbsCoef(layer), k_ref)
    # per-molecule getters keep the per-list path
    for m, name in zip(layer, ("co2", "h2o", "ch4")):
        assert rel_err(pyrad.getAbsCoef(m), z[name + ".abs_coef"]) <= RTOL

    # cross sections first, then a mutator, then the layer
    layer = build()
    assert np.array_equal(pyrad.getCrossSection(layer[0][1]), xs_ref[0][1])
    layer.changeTemperature(250)
    assert not layer[0][1].progressCrossSection
    k250 = np.array(pyrad.getAbsCoef(layer))
    assert rel_err(k250, k_ref250) <= 1e-13 and layer[0][1]._xs_deferred
    assert np.array_equal(pyrad.getCrossSection(layer[0][1]), xs_ref250)
    # a depth change redoes the transmittance from the resident absorption coefficient (no accumulate job)
    t1 = np.array(pyrad.getTransmittance(layer))
    layer.changeDepth(2.0 * layer.depth)
    t2 = np.array(pyrad.getTransmittance(layer))
    assert rel_err(t2, t1 ** 2, floor=1e-300) <= 1e-9 and np.array_equal(pyrad.getAbsCoef(layer), k250)


def test_atmosphere_transmission_merged_equals_per_list(pyrad):
    """Atmosphere.transmission through one merged accumulate job per layer + the fold over the absorption coefficients
    against the per-line-list route; layers' own getters afterwards find their arrays resident."""
    from pyrad_amd import settings
    lines = dict(co2=synthetic.make_lines(51, 800, 580, 720), h2o=synthetic.make_lines(52, 500, 580, 720))
    source(**lines)

    def build():
        pyrad.Layer.hasAtmosphere = False
        atm = pyrad.Atmosphere("col")
        for depth, T, P in ((1e4, 288, 1013.25), (2e4, 270, 700.0), (5e4, 240, 300.0), (1e5, 220, 80.0)):
            L = atm.addLayer(depth, T, P, 600, 700)
            L.addMolecule('co2', ppm=400)
            L.addMolecule('h2o', percentage=0.5)
        return atm

    settings.set_layer_step("per-list")
    try:
        ref_atm = build()
        ref = np.array(ref_atm.transmission(surfaceTemperature=288))
        ref_k = [np.array(pyrad.getAbsCoef(L)) for L in ref_atm]
    finally:
        settings.set_layer_step("merged")
    atm = build()
    got = np.array(atm.transmission(surfaceTemperature=288))
    assert rel_err(got, ref) <= 1e-12
    for L, k in zip(atm, ref_k):
        assert rel_err(pyrad.getAbsCoef(L), k) <= 1e-13
        assert rel_err(pyrad.getTransmittance(L), np.exp(-k * L.depth), floor=1e-300) <= 1e-9
    surf = atm[0].planck(300)
    assert rel_err(atm.transmission(surfaceSpectrum=surf), ref_atm.transmission(surfaceSpectrum=surf)) <= 1e-12
    atm[2].changeTemperature(250)                      # one layer due: only its job runs, the fold covers all
    ref_atm[2].changeTemperature(250)
    settings.set_layer_step("per-list")
    try:
        ref2 = np.array(ref_atm.transmission(surfaceTemperature=288))
    finally:
        settings.set_layer_step("merged")
    assert rel_err(atm.transmission(surfaceTemperature=288), ref2) <= 1e-12


def test_atmosphere_transmission_in_pieces_equals_one_pass(pyrad):
    """a column wide enough (100,001 points) for Atmosphere.transmission to fold it in four pieces, each piece's part of
    the outgoing spectrum downloaded beside the next piece (lbl_buffer_download_async): the same spectrum as the per-list
    route's one pass, with a surface temperature and with a surface spectrum; a second call reuses the resident buffers"""
    from pyrad_amd import settings
    lines = dict(co2=synthetic.make_lines(61, 3000, 590, 710), h2o=synthetic.make_lines(62, 2000, 590, 710))
    source(**lines)
    keep = settings.RES_MULTIPLIER
    settings.set_resolution_multiplier(0.1)               # base resolution 0.001 cm^-1

    def build():
        pyrad.Layer.hasAtmosphere = False
        atm = pyrad.Atmosphere("col")
        for depth, T, P in ((1e4, 288, 1013.25), (3e4, 260, 500.0), (8e4, 230, 120.0)):
            L = atm.addLayer(depth, T, P, 600, 700)
            L.addMolecule('co2', ppm=400)
            L.addMolecule('h2o', percentage=0.4)
        return atm

    try:
        settings.set_layer_step("per-list")
        try:
            ref_atm = build()
            ref = np.array(ref_atm.transmission(surfaceTemperature=288))
            surf = np.array(ref_atm[0].planck(300))
            ref_s = np.array(ref_atm.transmission(surfaceSpectrum=surf))
        finally:
            settings.set_layer_step("merged")
        atm = build()
        got = np.array(atm.transmission(surfaceTemperature=288))
        assert got.size >= (1 << 16) and got.size == ref.size
        assert np.all(np.isfinite(got)) and rel_err(got, ref) <= 1e-12
        assert rel_err(atm.transmission(surfaceSpectrum=surf), ref_s) <= 1e-12
        assert np.array_equal(atm.transmission(surfaceTemperature=288), got)
        k = np.array(pyrad.getAbsCoef(atm[1]))
        assert rel_err(pyrad.getTransmittance(atm[1]), np.exp(-k * atm[1].depth), floor=1e-300) <= 1e-9
    finally:
        settings.set_resolution_multiplier(keep)


def _many_isotopologue_source(molecule_ids, n_lines, lo, hi, seed=700):
    """A MemorySource holding a small line list for EVERY isotopologue of the given HITRAN molecule numbers (global
    isotopologue ids from the model's table, cls:951-1016), with a power-law partition sum and a molar mass per id."""
    from pyrad_amd import data, model
    src = data.MemorySource()
    spec = {}
    for m in molecule_ids:
        for k, gid in model.HITRAN_GLOBAL_ISO[m].items():
            lines = synthetic.make_lines(seed + gid, n_lines, lo, hi)
            q296, beta, molmass = 150.0 + 7.0 * gid, 1.0 + 0.01 * (gid % 50), 16.0 + 0.37 * gid
            t = np.arange(1, 1001, dtype=np.float64)
            q = {int(a): float(b) for a, b in zip(t, q296 * (t / 296.0) ** beta)}
            src.register(gid, lines, q, [gid, "M%d" % m, m, k, 1.0, q296, 1, molmass])
            spec[gid] = dict(lines=lines, q=q, q296=q296, molmass=molmass)
    data.set_source(src)
    return spec


@pytest.mark.parametrize("molecule_ids,n_lists", [((1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14), 57),
                                                   ((1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19), 74)])
def test_layer_of_any_size_against_the_oracle(pyrad, molecule_ids, n_lists):
    """The reference sums however many molecules and isotopologues a layer holds (cls:566-571, 707-712; addMolecule with
    isotopeDepth, cls:757-779, 446).  57 line lists: one merged accumulate job (the library takes 64 per job); 74: more
    than a merged job or a sweep's argument block holds - the per-line-list step, swept by the column-step kernel on a
    column of one layer.  Every point against the oracle's sum over 57 / 74 cross sections, through model.Layer; a column
    containing such a layer through Atmosphere.transmission (round-5 verdict, item 3c)."""
    from oracle import pyrad_oracle as orc
    from pyrad_amd import _native as nat, settings
    spec = _many_isotopologue_source(molecule_ids, 300, 590, 710)
    assert nat.limit("merged_lists_per_job") == 64 and nat.limit("arrays_per_layer") == 511

    def build(T, P, atmosphere=None):
        if atmosphere is None:
            pyrad.Layer.hasAtmosphere = False
            layer = pyrad.Layer(1000.0, T, P, 600, 700)
        else:
            layer = atmosphere.addLayer(1000.0, T, P, 600, 700)
        for j, m in enumerate(molecule_ids):
            layer.addMolecule(m, isotopeDepth=len(pyrad.HITRAN_GLOBAL_ISO[m]), ppm=50.0 + 10.0 * j)
        return layer

    def oracle_k(layer):
        g = orc.layer_grid(layer.P, 600, 700, 0.01, True)
        k = np.zeros(g["n_base"])
        for mol in layer:
            xs_m = np.zeros(g["n_base"])
            for iso in mol:
                s = spec[iso.globalIsoNumber]
                sel = orc.select_window(s["lines"], g["eff_min"], g["eff_max"])
                xs = orc.create_cross_section(sel, layer.T, layer.P, mol.concentration, s["molmass"],
                                              s["q"][int(layer.T)], s["q296"], g)[0]
                xs_m = xs_m + xs
            k = k + orc.abs_coef(xs_m, mol.concentration, layer.P, layer.T)
        return k

    layer = build(296, 1013.25)
    assert sum(len(m) for m in layer) == n_lists
    k_ref = oracle_k(layer)
    k = np.array(pyrad.getAbsCoef(layer))
    assert rel_err(k, k_ref) <= RTOL
    merged = n_lists <= 64
    assert all(iso._xs_deferred == merged for m in layer for iso in m)          # which route ran
    assert rel_err(pyrad.getTransmittance(layer), np.exp(-k_ref * layer.depth), floor=1e-300) <= 1e-9
    # the per-line-list route on the same layer (a sweep over 57 arrays fits its argument block; 74 go through the column-step kernel)
    settings.set_layer_step("per-list")
    try:
        other = build(296, 1013.25)
        assert rel_err(pyrad.getAbsCoef(other), k_ref) <= RTOL
        xs_layer = np.array(pyrad.getCrossSection(other))                      # sum of 14 / 19 molecule sums of up to 12 arrays
        assert xs_layer.shape == k_ref.shape and np.all(np.isfinite(xs_layer))
    finally:
        settings.set_layer_step("merged")
    # a column with such a layer between two ordinary ones
    pyrad.Layer.hasAtmosphere = False
    atm = pyrad.Atmosphere("col")
    small = []
    for T, P in ((288, 1013.25), (250, 400.0)):
        L = atm.addLayer(2000.0, T, P, 600, 700)
        L.addMolecule(2, ppm=400)
        small.append(L)
    big = build(270, 700.0, atmosphere=atm)
    order = [small[0], big, small[1]]
    atm[:] = order
    got = np.array(atm.transmission(surfaceTemperature=288))
    xa = orc.x_axis(600, 700, 0.01)
    I = orc.planckWavenumber(xa, 288)
    for L in order:
        tr = orc.transmittance(oracle_k(L), L.depth)
        I = orc.transmission(tr, I, orc.planckWavenumber(xa, L.T))
    assert rel_err(got, I) <= RTOL


def test_an_installed_cross_section_is_what_the_layer_sums(pyrad):
    """advisor, round 5: ``iso.crossSection = array`` with progressCrossSection left set is what the reference's getters use
    (getCrossSection does not recompute, cls:32-35); when a SIBLING line list is due, the merged layer step (all lines of
    the layer) would silently recompute the layer from the installed isotopologue's LINES.  A layer holding an installed
    array takes the per-line-list route: merged setting and per-list setting agree, and both show the installed array."""
    from pyrad_amd import settings
    z = load_golden("G6_composition")
    source(co2=unpack_lines(z, "co2.lines"), co2_636=unpack_lines(z, "co2_636.lines"), h2o=unpack_lines(z, "h2o.lines"))

    def run(step):
        settings.set_layer_step(step)
        try:
            pyrad.Layer.hasAtmosphere = False
            layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), 1000, 1040)
            layer.addMolecule('co2', isotopeDepth=2, ppm=400)
            layer.addMolecule('h2o', **{'%': 1.5})
            k0 = np.array(pyrad.getAbsCoef(layer))
            xs0 = np.array(pyrad.getCrossSection(layer[0][0]))
            layer[0][0].crossSection = 3.0 * xs0                      # somebody's own array; the flag stays set
            assert layer[0][0].progressCrossSection
            pyrad.resetCrossSection(layer[0][1])                      # a sibling is due
            k1 = np.array(pyrad.getAbsCoef(layer))
            assert np.array_equal(layer[0][0].crossSection, 3.0 * xs0)
            layer.changeTemperature(250)                              # resets everything: the installed array is gone (cls:38-45)
            k2 = np.array(pyrad.getAbsCoef(layer))
            return k0, k1, k2, xs0
        finally:
            settings.set_layer_step("merged")

    a, b = run("merged"), run("per-list")
    for x, y in zip(a, b):
        assert rel_err(x, y) <= 1e-13
    k0, k1, k2, xs0 = a
    f = 400e-6 * float(z["P"]) / 1e4 / 1.38064852E-23 / int(z["T"])
    assert rel_err(k1 - k0, 2.0 * xs0 * f, floor=float(np.max(k0)) * 1e-3) <= 1e-9      # the layer gained exactly 2 x that cross section
    assert rel_err(k2, k0) > 1e-3


def test_the_reference_rounding_chain_takes_the_per_list_route(pyrad):
    """advisor, round 5: "sweep_ieee_divisions" 1 (the reference's own chain of correctly rounded divisions) used to be silently
    ignored by the merged entry points, which the object model takes by default.  Now the library refuses it there
    (LBL_ERR_BAD_ARG) and the model, seeing the option on its context, takes the per-line-list route, which honours it: the
    absorption coefficient is then bit for bit NumPy's expression on the device's cross sections."""
    from pyrad_amd import _native as nat
    z = load_golden("G6_composition")
    source(co2=unpack_lines(z, "co2.lines"), co2_636=unpack_lines(z, "co2_636.lines"), h2o=unpack_lines(z, "h2o.lines"))
    pyrad.Layer.hasAtmosphere = False
    layer = pyrad.Layer(float(z["depth"]), int(z["T"]), float(z["P"]), 1000, 1040)
    layer.addMolecule('co2', isotopeDepth=2, ppm=400)
    layer.addMolecule('h2o', **{'%': 1.5})
    ctx = pyrad._ctx()
    ctx.set_option("sweep_ieee_divisions", 1)
    try:
        k = np.array(pyrad.getAbsCoef(layer))
        assert not any(iso._xs_deferred for m in layer for iso in m)          # the per-list route ran: the cross sections exist
        xs = [sum(np.array(pyrad.getCrossSection(iso)) for iso in m) for m in layer]
        want = sum(x * m.concentration * layer.P / 1E4 / 1.38064852E-23 / layer.T for x, m in zip(xs, layer))
        assert np.array_equal(k, want)
        g = pyrad._engine.native_grid(layer._grid())
        flat = [iso for m in layer for iso in m]
        with pytest.raises(nat.LblError, match="default arithmetic"):
            ctx.layer_merged_step_dev([i._device_lines(ctx) for i in flat], [pyrad._iso_params(i) for i in flat], g, [0, 0, 1],
                                      [m.concentration for m in layer], layer.depth, abs_coef=ctx.buffer(g.n_base))
    finally:
        ctx.set_option("sweep_ieee_divisions", 0)


def test_resident_column_follows_every_kind_of_change(pyrad):
    """Atmosphere.transmission through the resident column handle (lbl_column, round 6) against a freshly built atmosphere
    through the general route, after each kind of change between calls: temperature, pressure (new windows), range, depth,
    concentration, a molecule swapped for another one, a layer added, the surface given as a spectrum - and unchanged calls."""
    lines = dict(co2=synthetic.make_lines(71, 900, 580, 720), h2o=synthetic.make_lines(72, 600, 580, 720), ch4=synthetic.make_lines(73, 300, 580, 720))
    source(**lines)
    spec = [(1e4, 288, 1013.25), (2e4, 270, 700.0), (5e4, 240, 300.0)]

    def build(mods):
        pyrad.Layer.hasAtmosphere = False
        atm = pyrad.Atmosphere("col")
        for depth, T, P in spec:
            L = atm.addLayer(depth, T, P, 600, 700)
            L.addMolecule('co2', ppm=400)
            L.addMolecule('h2o', percentage=0.5)
        for f in mods:
            f(atm)
        return atm

    mods = []
    live = build(mods)
    first = np.array(live.transmission(surfaceTemperature=288))
    assert "_column_fast" in live.__dict__                                   # the handle exists after the first call
    assert np.array_equal(live.transmission(surfaceTemperature=288), first)  # unchanged: nothing is due, the fold runs

    def swap_molecule(a):
        gone = a[2].pop(1)
        a[2].addMolecule(6, ppm=1.8)

    steps = [lambda a: a[1].changeTemperature(255), lambda a: a[0].changePressure(900.0), lambda a: a[2].changeDepth(7e4),
             lambda a: a[1][0].setPPM(420), swap_molecule, lambda a: [L.changeRange(610, 690) for L in a],
             lambda a: a.addLayer(8e4, 220, 80.0, 610, 690).addMolecule('co2', ppm=400)]
    for i, f in enumerate(steps):
        f(live)
        mods.append(f)
        got = np.array(live.transmission(surfaceTemperature=288))
        fresh = build(mods)
        fresh.__dict__["_column_fast"] = None
        want = np.array(fresh.transmission(surfaceTemperature=288))
        assert got.shape == want.shape and rel_err(got, want) <= 1e-13, i
        surf = np.array(live[0].planck(300))
        assert rel_err(live.transmission(surfaceSpectrum=surf), fresh.transmission(surfaceSpectrum=surf)) <= 1e-13, i
        assert np.array_equal(live.transmission(surfaceTemperature=288), got), i
