"""CPU: the reader of PyRad's data/ tree (pyrad_amd.data.PyradDataDir) against G9 - what the
reference's OWN openReturnLines / gatherData / readHitranOnlineFile / readQFile / readMolParams
(ut:90-101, 173-189, 421-477, executed unmodified in the build container) returned for a tree with
duplicate wavenumbers within and across segments, lines exactly on the window edges, '#' headers, a
NULL_TAG segment, unsorted rows and windows that start mid-segment; and the oracle's line survey
(cls:409-428, 589-594, 691-696) against G10."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden, unpack_lines
from pyrad_amd import data, synthetic
from oracle import pyrad_oracle as orc

FIELDS = (("intensity", "sw"), ("einsteinA", "a"), ("airHalfWidth", "gamma_air"), ("selfHalfWidth", "gamma_self"),
          ("lowerEnergy", "elower"), ("tempExponent", "n_air"), ("pressureShift", "delta_air"))


def write_data_tree(z, root):
    """Materialise the data/ tree stored in G9 (the bytes the reference's readers were given)."""
    for i, key in enumerate(json.loads(str(z["tree_json"]))):
        full = os.path.join(root, key)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        with open(full, "wb") as f:
            f.write(z["tree.%d" % i].tobytes())


@pytest.fixture()
def tree(tmp_path):
    z = load_golden("G9_data_dir")
    write_data_tree(z, str(tmp_path))
    return z, data.PyradDataDir(str(tmp_path))


def test_gather_data_matches_the_reference_reader(tree):
    z, src = tree
    queries = json.loads(str(z["queries_json"]))
    assert len(queries) == 7
    for qi, (iso, lo, hi) in enumerate(queries):
        if "q%d.raises" % qi in z.files:              # segments that are not in the cache: the reference downloads
            with pytest.raises(FileNotFoundError):
                src.gatherData(iso, lo, hi)
            continue
        got = src.gatherData(iso, lo, hi)
        ref_nu = z["q%d.nu" % qi]
        # the reference returns a dict in file order (later duplicates replace the value, ut:447 / dict.update
        # ut:187); the device wants the list sorted by wavenumber: same lines, same values
        order = np.argsort(ref_nu, kind="stable")
        assert np.array_equal(got["nu"], ref_nu[order]), qi
        assert len(np.unique(got["nu"])) == len(got["nu"])
        for ref_name, name in FIELDS:
            assert np.array_equal(got[name], z["q%d.%s" % (qi, ref_name)][order]), (qi, name)
    # the edge cases are really in there (reference behaviour, not this build's choice)
    nu0, s0 = z["q0.nu"], z["q0.intensity"]
    assert 595.0 not in nu0 and 705.0 not in nu0 and 595.000001 in nu0 and 704.999999 in nu0      # strict bounds
    assert s0[nu0 == 600.0][0] == 5e-20            # duplicate across segments: the later file wins
    assert s0[nu0 == 650.123456][0] == 7e-20       # duplicate within a file: the later row wins
    assert s0[nu0 == 700.0][0] == 9e-20
    assert not np.all(np.diff(nu0) > 0)            # the shuffled file stayed shuffled in the reference's dict
    assert len(z["q5.nu"]) == 0                    # NULL_TAG segment: no lines, no error


def test_q_file_and_params_match_the_reference_reader(tree):
    z, src = tree
    for iso in (7, 1):
        q = src.getQData(iso)
        assert list(q.keys()) == list(z["qfile.%d.T" % iso]) and list(q.values()) == list(z["qfile.%d.Q" % iso])
        assert src.readMolParams(iso) == json.loads(str(z["params.%d_json" % iso]))
    assert data.PyradDataDir.segments(595.0, 705.0) == [500, 600, 700]
    assert data.PyradDataDir.segments(650.5, 660.25) == [600]


def test_reader_is_as_strict_as_the_reference(tmp_path):
    """A malformed row is an IndexError / ValueError in readHitranOnlineFile (ut:434-446) and readQFile
    (ut:457-460); it is one here."""
    root = str(tmp_path)
    lines = synthetic.make_lines(9, 20, 600, 700)
    data.PyradDataDir.write_tree(root, 7, lines, synthetic.q_table("co2", 300), synthetic.mol_params("co2"), 2, 1)
    src = data.PyradDataDir(root)
    assert len(src.gatherData(7, 600, 700)["nu"]) == 20
    with open(os.path.join(root, "7", "600.pyr"), "a") as f:
        f.write("\n")
    with pytest.raises(IndexError):
        src.gatherData(7, 600, 700)
    with open(os.path.join(root, "7", "q7.txt"), "a") as f:
        f.write("\n")
    with pytest.raises(IndexError):
        src.getQData(7)


def test_layers_through_the_data_dir_match_the_oracle(tree):
    """The layers G9 ran through the reference's classes ON TOP of its readers: the oracle, fed by
    PyradDataDir, reproduces absorption coefficient, transmittance and radiance."""
    z, src = tree
    for tag in json.loads(str(z["layer_cases_json"])):
        spec = json.loads(str(z["%s.spec_json" % tag]))
        g = orc.layer_grid(spec["P"], spec["rmin"], spec["rmax"], .01, True)
        k = np.zeros(g["n_base"])
        for iso, species, conc, key in ((7, "co2", orc.concentration(ppm=400), "co2"), (1, "h2o", orc.concentration(percentage=1.2), "h2o")):
            lines = src.gatherData(iso, g["eff_min"], g["eff_max"])
            p = src.readMolParams(iso)
            xs, _ = orc.create_cross_section(lines, spec["T"], spec["P"], conc, p[7], src.getQData(iso)[spec["T"]], p[5], g)
            assert np.max(np.abs(xs - z["%s.%s.xsec" % (tag, key)]) / np.maximum(z["%s.%s.xsec" % (tag, key)], 1e-300)) <= 1e-12
            k = k + orc.abs_coef(xs, conc, spec["P"], spec["T"])
        assert np.max(np.abs(k - z["%s.abs_coef" % tag]) / z["%s.abs_coef" % tag]) <= 1e-12
        tr = orc.transmittance(k, spec["depth"])
        assert np.max(np.abs(tr - z["%s.transmittance" % tag]) / z["%s.transmittance" % tag]) <= 1e-12
        xa = orc.x_axis(spec["rmin"], spec["rmax"], .01)
        I = orc.transmission(tr, orc.planckWavenumber(xa, 288), orc.planckWavenumber(xa, spec["T"]))
        assert np.max(np.abs(I - z["%s.transmission" % tag]) / z["%s.transmission" % tag]) <= 1e-12


def test_oracle_line_survey_matches_g10():
    z = load_golden("G10_line_survey")
    for tag in json.loads(str(z["cases_json"])):
        spec = json.loads(str(z["%s.spec_json" % tag]))
        res = float(z["%s.resolution" % tag])
        g = orc.layer_grid(spec["P"], spec["range_min"], spec["range_max"], .01, True)
        assert g["resolution"] == res
        layer_sum = np.zeros(g["n_base"])
        for mi, m in enumerate(spec["molecules"]):
            mol_sum = np.zeros(g["n_base"])
            for ii in range(m["isotope_depth"]):
                lines = unpack_lines(z, "%s.mol%d.lines%s" % (tag, mi, "2" if ii else ""))
                sel = orc.select_window(lines, g["eff_min"], g["eff_max"])     # what gatherData hands to getData
                s = orc.line_survey(sel["nu"], sel["sw"], spec["range_min"], spec["range_max"], res, .01)
                assert np.array_equal(s, z["%s.mol%d.iso%d" % (tag, mi, ii)]), (tag, mi, ii)
                mol_sum += s
            assert np.array_equal(mol_sum, z["%s.mol%d" % (tag, mi)])
            layer_sum += mol_sum
        assert np.array_equal(layer_sum, z["%s.layer" % tag])
    # the res != BASE quirk (cls:416 vs 423): 0.1 cm^-1 bins in an array of 0.01 cm^-1 length, so the 20 cm^-1
    # range fills bins 0..199 and lines from the +50 cm^-1 window margin land in bins 200..699 of the 2000
    s3 = z["S3.layer"]
    assert s3.size == 2000 and np.count_nonzero(s3[200:700]) > 0 and np.count_nonzero(s3[700:]) == 0


def test_cached_reader_equals_the_uncached_one(tree, tmp_path, monkeypatch):
    """PyradDataDir keeps parsed segments (round 4: re-windowing re-asks for the same files) and hands out slices of one
    sorted, duplicate-free list per isotopologue while its segments are well formed; with cache=False it reads and
    parses row by row on every call, like the reference.  Same results on the G9 tree (which holds a row filed in the
    wrong segment: that isotopologue stays on the row-by-row path) and on a well-formed tree; files are read once and
    re-read when they change."""
    z, cached = tree
    plain = data.PyradDataDir(cached.root, cache=False)
    for qi, (iso, lo, hi) in enumerate(json.loads(str(z["queries_json"]))):
        if "q%d.raises" % qi in z.files:
            continue
        a, b = cached.gatherData(iso, lo, hi), plain.gatherData(iso, lo, hi)
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a), qi
    # a well-formed tree: duplicates within a file and across windows, unsorted rows
    root = str(tmp_path / "clean")
    lines = synthetic.make_lines(11, 3000, 480.0, 920.0, decimals=2)          # two decimals: duplicated wavenumbers
    rng = np.random.default_rng(3)
    shuffled = {k: v[rng.permutation(len(lines["nu"]))] for k, v in lines.items()}
    shuffled["a"] = np.zeros_like(shuffled["nu"])
    data.PyradDataDir.write_tree(root, 7, shuffled, {296: 286.0}, synthetic.mol_params("co2"))
    cached, plain = data.PyradDataDir(root), data.PyradDataDir(root, cache=False)
    reads = []
    real_rows = data.PyradDataDir._rows
    monkeypatch.setattr(data.PyradDataDir, "_rows", staticmethod(lambda path: (reads.append(path), real_rows(path))[1]))
    for lo, hi in ((595.0, 705.0), (600.0, 700.0), (612.34, 612.36), (520.0, 880.0), (595.0, 705.0), (640.0, 641.0)):
        a, b = cached.gatherData(7, lo, hi), plain.gatherData(7, lo, hi)
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a), (lo, hi)
        hit = data.master_slice(a, ("nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air"))
        assert a["nu"].size == 0 or (hit is not None and hit[2] == a["nu"].size)            # a slice of the isotopologue's list
    assert len(set(reads)) == 4                                                          # segments 500 .. 800
    assert len(reads) == 4 + sum(len(data.PyradDataDir.segments(lo, hi)) for lo, hi in
                                            ((595.0, 705.0), (600.0, 700.0), (612.34, 612.36), (520.0, 880.0), (595.0, 705.0), (640.0, 641.0)))
    # a file that changes on disk is read again
    path = os.path.join(root, "7", "600.pyr")
    rows = open(path).read().splitlines()
    extra = rows[0].split(","); extra[2] = "650.555555"; extra[3] = "1.25e-19"
    with open(path, "w") as f:
        f.write("\n".join(rows + [",".join(extra)]) + "\n")
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 10_000_000))
    a, b = cached.gatherData(7, 640.0, 660.0), plain.gatherData(7, 640.0, 660.0)
    assert 650.555555 in a["nu"] and all(np.array_equal(a[k], b[k]) for k in a)
    # an unparsable column only matters for the rows a window reads (ut:434-446), cached or not
    rows = open(path).read().splitlines()
    bad = rows[1].split(","); bad[2] = "699.987654"; bad[5] = "not-a-number"
    with open(path, "w") as f:
        f.write("\n".join(rows + [",".join(bad)]) + "\n")
    os.utime(path, ns=(os.stat(path).st_atime_ns, os.stat(path).st_mtime_ns + 20_000_000))
    for src in (cached, plain):
        assert 699.987654 not in src.gatherData(7, 640.0, 660.0)["nu"]
        with pytest.raises(ValueError):
            src.gatherData(7, 690.0, 700.0)
