"""CPU: host-side logic of the build — layer grid scalars, data readers in PyRad's on-disk
formats, the dirty-flag protocol wiring, tables and unit converters."""
import os

import numpy as np
import pytest

from conftest import load_golden, unpack_lines
from pyrad_amd import data, engine, model, settings, synthetic
from oracle import pyrad_oracle as orc


def test_layer_grid_matches_oracle_and_goldens():
    z = load_golden("G3_pressure_ladder")
    for j, P in enumerate(z["P_list"]):
        g = engine.layer_grid(float(P), 640, 660, .01, True)
        o = orc.layer_grid(float(P), 640, 660, .01, True)
        for key in ("dfc", "eff_min", "eff_max", "resolution", "n_base", "n_work", "W"):
            assert g[key] == o[key], key
        assert (g["W"], g["n_work"]) == (int(z["P%d.W" % j]), int(z["P%d.n_work" % j]))
    g = engine.layer_grid(1013.25, 500, 900, .001, False)
    assert (g["n_base"], g["W"], g["resolution"]) == (400000, 5000, .001)
    assert np.array_equal(engine.x_axis(600, 700, .01), load_golden("G1_c1_cell")["x_axis"])


def test_eval_count_matches_oracle():
    cfg = synthetic.config_c1(n_lines=500)
    lines = cfg["molecules"][0]["lines"]
    g = orc.layer_grid(1013.25, 600, 700, .01, True)
    sel = orc.select_window(lines, g["eff_min"], g["eff_max"])
    lq = orc.line_quantities(sel, 296, 1013.25, 4e-4, 43.98983, 600, .01)
    total = orc.eval_count(lq["index"], g["W"], g["n_work"])
    assert engine.eval_count(sel["nu"], 600, .01, g["W"], g["n_work"]) == total
    parts = 0
    for rank in range(3):
        S, first, count = engine.shard_bounds(g["n_work"], 3, rank)
        parts += engine.eval_count(sel["nu"], 600, .01, g["W"], g["n_work"], (first, count))
    assert parts == total


def test_concentration_setters_and_converters_match_golden():
    z = load_golden("G0_functions")
    got = [model.concentration_from_kwargs(ppm=400.0), model.concentration_from_kwargs(ppb=1.0),
           model.concentration_from_kwargs(ppb=1800.0), model.concentration_from_kwargs(**{"%": 1.0}),
           model.concentration_from_kwargs(concentration=0.0004)]
    assert np.array_equal(np.array(got), z["conc_values"])
    conv = [model.convertLength(2.0, "m"), model.convertLength(2.0, "ft"), model.convertLength(2.0, "in"),
            model.convertPressure(2.0, "atm"), model.convertPressure(2.0, "bar"), model.convertPressure(2.0, "pa"),
            model.convertRange(15.0, "um"), model.convertTemperature(15.0, "C"), model.convertTemperature(59.0, "F")]
    assert np.array_equal(np.array(conv), z["convert"])
    assert model.convertTemperature(15.0, "C") == 288          # +273, not 273.15 (cls:154)


def test_tables():
    assert model.MOLECULE_ID['co2'] == 2 and model.MOLECULE_ID['cocl2'] == 49 and len(model.MOLECULE_ID) == 49
    assert model.getGlobalIsotope(2, 3) == [7, 8, 9]
    assert model.HITRAN_GLOBAL_ISO[2][12] == 122          # the cls:952 copy of the table
    assert model.HITRAN_GLOBAL_ISO[16] == {1: 19, 2: 11, 3: 111, 4: 112}
    assert model.getGlobalIsotope(6, 1) == [32] and model.getGlobalIsotope(3, 1) == [16]


def test_pyrad_data_dir_round_trip(tmp_path):
    """Reader of PyRad's on-disk cache: column order, strict bounds, last-wins duplicates,
    '#' headers, the NULL_TAG sentinel, 100 cm^-1 segment names."""
    lines = synthetic.make_lines(7, 400, 580, 730)
    root = str(tmp_path / "data")
    data.PyradDataDir.write_tree(root, 7, lines, synthetic.q_table("co2", 400), synthetic.mol_params("co2"), 2, 1)
    assert sorted(f for f in os.listdir(root + "/7") if f.endswith(".pyr")) == ["500.pyr", "600.pyr", "700.pyr", "params.pyr"]
    src = data.PyradDataDir(root)
    assert src.readMolParams(7) == [7, "CO2", 2, 1, 1.0, 286.09, 1, 43.98983]
    assert src.getQData(7)[296] == synthetic.q_table("co2", 400)[296]
    got = src.gatherData(7, 595.0, 705.0)
    m = (lines["nu"] > 595.0) & (lines["nu"] < 705.0)
    for k in ("nu", "sw", "a", "elower", "gamma_air", "gamma_self", "delta_air", "n_air"):
        assert np.array_equal(got[k], lines[k][m]), k
    assert data.PyradDataDir.segments(595.0, 705.0) == [500, 600, 700]
    # strict bounds (ut:437-438)
    nu0 = float(lines["nu"][m][0])
    assert src.gatherData(7, nu0, 705.0)["nu"][0] > nu0
    # duplicate wavenumber: the later row wins (ut:447)
    with open(root + "/7/600.pyr", "a") as f:
        f.write("2,1,%r,9.9e-20,1.0,10.0,0.07,0.09,-0.001,0.7\n" % float(lines["nu"][m][5]))
    got2 = src.gatherData(7, 595.0, 705.0)
    assert len(got2["nu"]) == m.sum() and got2["sw"][5] == 9.9e-20
    # NULL_TAG segment = empty; missing segment = loud error (the reference would download)
    with open(root + "/7/700.pyr", "w") as f:
        f.write(data.NULL_TAG + "\n")
    assert src.gatherData(7, 595.0, 705.0)["nu"].max() < 700
    os.remove(root + "/7/500.pyr")
    with pytest.raises(FileNotFoundError):
        src.gatherData(7, 595.0, 705.0)


def test_memory_source_dedupe_and_window():
    src = data.MemorySource()
    lines = synthetic.make_lines(9, 50, 600, 610)
    dup = {k: np.concatenate([v, v[3:4]]) for k, v in lines.items()}
    dup["sw"][-1] = 1.23e-19
    src.register(7, dup, {296: 286.09}, synthetic.mol_params("co2"))
    got = src.gatherData(7, 600, 610)
    assert len(got["nu"]) == 50 and got["sw"][3] == 1.23e-19
    assert np.all(np.diff(got["nu"]) > 0)


def test_shard_bounds_cover_grid():
    for n in (1, 7, 400000, 2400001):
        for world in (1, 2, 3, 8):
            S = None
            covered = 0
            for r in range(world):
                s, first, count = engine.shard_bounds(n, world, r)
                S = s
                assert first == min(r * s, n) and 0 <= count <= s
                covered += count
            assert covered == n and S * world >= n


def test_pair_split_matches_a_brute_force_classification():
    """dist.pair_split (bench.py's direct / series bookkeeping): classes of (line, span of 256 points) pairs as
    lbl_api.hip's group_schedule tabulates them, against a loop over spans and lines; evals add up to eval_count."""
    import numpy as np
    from pyrad_amd import dist
    rng = np.random.default_rng(5)
    for H, n, first, count in ((4998, 9000, 0, None), (700, 5000, 1024, 2048), (300, 4000, 0, None), (1500, 3000, 256, 1500)):
        c = np.sort(rng.integers(-H - 50, n + H + 50, 400))
        got = dist.pair_split(c, H, n, first, count)
        cnt = n - first if count is None else count
        pairs = far = far_evals = evals = 0
        for lo in range(first, first + cnt, 256):
            hi = min(lo + 256, first + cnt) - 1
            xc2 = 2 * lo + 255                                  # twice the span centre lo + 127.5
            for ci in c:
                if ci + H < lo or ci - H > hi:
                    continue
                pairs += 1
                covers = ci - H <= lo and ci + H >= hi
                # far: centre index <= lo + 127 - 512 or >= lo + 128 + 512 (lbl_kernels.hip: FF_FAR half-spans of 128)
                if H >= 640 and covers and (ci <= lo + 127 - 512 or ci >= lo + 128 + 512):
                    far += 1
                    far_evals += hi - lo + 1
        for ci in c:
            evals += max(min(ci + H, first + cnt - 1) - max(ci - H, first) + 1, 0)
        assert got["pairs"] == pairs and got["pairs_series"] == far and got["pairs_direct"] == pairs - far, (H, n)
        assert got["evals"] == evals and got["evals_series"] == far_evals and got["evals_direct"] == evals - far_evals


def test_memory_source_windows_are_slices_with_the_reference_semantics():
    """MemorySource.gatherData: strict bounds (ut:437-438), duplicated wavenumbers keep the LAST row's values at the
    FIRST row's position (ut:447) - collapsed once at registration - and every window is a zero-copy slice of the
    registered list that data.master_slice recognises (what lets the device hand out views instead of uploads)."""
    from pyrad_amd import data, synthetic
    rng = np.random.default_rng(7)
    lines = synthetic.make_lines(5, 400, 600.0, 700.0, decimals=1)             # one decimal: many duplicated wavenumbers
    order = rng.permutation(len(lines["nu"]))
    shuffled = {k: v[order] for k, v in lines.items()}
    src = data.MemorySource()
    src.register(7, shuffled, {296: 286.0}, synthetic.mol_params("co2"))

    def reference(lo, hi):                       # the pre-round-4 implementation: mask, then collapse
        o = np.argsort(shuffled["nu"], kind="stable")
        full = {k: np.asarray(v, dtype=np.float64)[o] for k, v in shuffled.items()}
        if "a" not in full:
            full["a"] = np.zeros_like(full["nu"])
        m = (full["nu"] > lo) & (full["nu"] < hi)
        return data._dedupe_last_wins({k: v[m] for k, v in full.items()})

    nu_all = np.unique(lines["nu"])
    for lo, hi in ((599.0, 701.0), (nu_all[3], nu_all[40]), (650.0, 650.0), (nu_all[10], nu_all[10]), (640.05, 640.15), (800.0, 900.0)):
        got, want = src.gatherData(7, lo, hi), reference(lo, hi)
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), (k, lo, hi)
        if got["nu"].size:
            assert got["nu"][0] > lo and got["nu"][-1] < hi and np.all(np.diff(got["nu"]) > 0)
        hit = data.master_slice(got, ("nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air"))
        if got["nu"].size:
            master, first, count = hit
            assert count == got["nu"].size and np.array_equal(master["nu"][first:first + count], got["nu"])
            assert got["nu"].base is master["nu"]                                  # no copy
    # anything that is not such a slice is not recognised
    sel = src.gatherData(7, 610.0, 690.0)
    assert data.master_slice({k: v.copy() for k, v in sel.items()}, ("nu", "sw")) is None
    assert data.master_slice(dict(sel, sw=sel["sw"][::-1].copy()), ("nu", "sw")) is None
    assert data.master_slice(dict(sel, sw=src.gatherData(7, 620.0, 690.0)["sw"]), ("nu", "sw")) is None
    full = src.gatherData(7, 0.0, 1e9)
    n = sel["nu"].size
    first = int(np.searchsorted(full["nu"], sel["nu"][0]))
    strided = full["sw"].base[first:first + 2 * n:2] if full["sw"].base is not None else None      # same start, same length, every other element
    if strided is not None and strided.size == n:
        assert data.master_slice(dict(sel, sw=strided), ("nu", "sw")) is None
