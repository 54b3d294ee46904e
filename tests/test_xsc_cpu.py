"""CPU: measured cross-section ("xsc") molecules — the file-name parser and reader (ut:611-715),
mergeArray (cls:165-233) and the oracle's restatement, all against G8 (captured from the
reference's own functions)."""
import json

import numpy as np
import pytest

from conftest import load_golden, write_xsc_tree
from pyrad_amd import data, model
from oracle import pyrad_oracle as orc


@pytest.fixture(scope="module")
def g8():
    return load_golden("G8_xsc")


def test_file_name_properties(g8):
    names = json.loads(str(g8["names_json"]))
    parsed = json.loads(str(g8["parsed_json"]))
    for name, want in zip(names, parsed):
        if "raises" in want:
            with pytest.raises(AttributeError):
                data.parseXscFileName(name)
        else:
            assert data.parseXscFileName(name) == want, name


def test_reader_and_exotic_table(g8, tmp_path):
    write_xsc_tree(g8, str(tmp_path))
    src = data.XscDir(str(tmp_path))
    for i, key in enumerate(json.loads(str(g8["tree_json"]))):
        mol, fn = key.split("/")
        got = src.processXscFile(mol, fn)
        assert np.array_equal(got["wavenumber"], g8["read.%d.wavenumber" % i])
        assert np.array_equal(got["intensity"], g8["read.%d.intensity" % i])
        assert got["res"] == float(g8["read.%d.res" % i])
    assert src.returnXscTemperaturePressureValues() == json.loads(str(g8["exotic_ids_json"]))
    try:
        data.set_xsc_source(src)
        assert set(model.EXOTIC_IDS) == {"CFC11", "CFC12", "CFC113"}
    finally:
        data.set_xsc_source(None)
    assert model.EXOTIC_IDS == {}

def test_missing_file_is_loud(tmp_path):
    src = data.XscDir(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        src.processXscFile("CFC11", "CFC11_296.0K-760.0Torr_620.0-680.0_0.01_air_12_34.txt")
    with pytest.raises(RuntimeError):
        data.get_xsc_source()


@pytest.mark.parametrize("merge", [model.mergeArray, orc.merge_array], ids=["model", "oracle"])
def test_merge_array_cases(g8, merge):
    for case in json.loads(str(g8["merge_cases_json"])):
        args = [g8["merge.%s.%s" % (case, k)] for k in ("newX", "oldX", "oldY")]
        if "merge.%s.raises" % case in g8.files:
            with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[str(g8["merge.%s.raises" % case])]):
                merge(*args)
        else:
            out = np.asarray(merge(*args), dtype=np.float64)
            assert np.array_equal(out, g8["merge.%s.out" % case]), case
    # list inputs are taken as they are (cls:166-173)
    want = g8["merge.inside_new.out"]
    got = model.mergeArray(*[g8["merge.inside_new.%s" % k].tolist() for k in ("newX", "oldX", "oldY")])
    assert np.array_equal(np.asarray(got, dtype=np.float64), want)


def test_oracle_xsc_cross_section(g8):
    axis = orc.x_axis(600, 700, .01)
    tree = json.loads(str(g8["tree_json"]))
    for tag in json.loads(str(g8["layer_cases_json"])):
        spec = json.loads(str(g8["%s.spec_json" % tag]))
        fn = spec["file"]
        if isinstance(fn, int):
            fn = [k.split("/")[1] for k in tree if k.startswith(spec["mol"] + "/")][fn]
        i = tree.index("%s/%s" % (spec["mol"], fn))
        p = data.parseXscFileName(fn)
        lo, hi = (float(v) for v in p["RANGE"].split("-"))
        xs = orc.xsc_cross_section(axis, lo, hi, float(p["RES"]), g8["read.%d.wavenumber" % i],
                                   g8["read.%d.intensity" % i])
        assert np.array_equal(np.asarray(xs, dtype=np.float64), g8["%s.mol_xsec" % tag]), tag
        T, P = orc.xsc_layer_state(p["TEMP"], p["PRESSURE"])
        assert (T, P) == (int(g8["%s.layer_T" % tag]), float(g8["%s.layer_P" % tag]))
        k = orc.abs_coef(g8["%s.mol_xsec" % tag], orc.concentration(**spec["conc"]), P, T)
        assert np.array_equal(k, g8["%s.mol_abs_coef" % tag])
    assert np.array_equal(np.interp(np.arange(590.0, 710.0, .01), g8["read.2.wavenumber"], g8["read.2.intensity"]),
                          g8["interp.out"])
