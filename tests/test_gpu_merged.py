"""GPU: merged layer jobs (ABI 4: lbl_layer_merged_step_dev, lbl_layers_merged_accumulate_dev, lbl_column_fold_dev) - ONE
accumulate job per layer over its merged, factor-weighted line lists, the absorption coefficient of
pyradClasses.py:707-712 accumulated directly - against the per-line-list path on the same resident inputs, over the launch
shapes the library routes differently (far-field kernel, all-direct kernel with line split, skewed-range kernel, regrid,
shards), and the error behaviour of the new entry points.  Every grid point against the C oracle at full size:
tests/test_gpu_whole_spectrum.py."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err
from pyrad_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrad_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def mols_of(cfg):
    from pyrad_amd.model import concentration_from_kwargs
    out = []
    for mol in cfg["molecules"]:
        sp = synthetic.SPECIES[mol["species"]]
        out.append(dict(conc=concentration_from_kwargs(**mol["conc"]),
                        isotopologues=[dict(lines=mol["lines"], molmass=sp["molmass"],
                                            q_T=synthetic.q_value(mol["species"], cfg["T"]), q296=sp["q296"])]))
    return out


def three_molecule_cell(n_lines, rmin, rmax, base, P, T=280, dynamic=False, seed=7):
    mk = lambda s: synthetic.make_lines(seed + s, n_lines, max(rmin - 30, 0.5), rmax + 30)
    return dict(depth=50.0, T=T, P=P, range_min=rmin, range_max=rmax, base_resolution=base, dynamic_resolution=dynamic,
                molecules=[dict(species="co2", conc=dict(ppm=400), lines=mk(0)), dict(species="h2o", conc={"%": 1.0}, lines=mk(1)),
                           dict(species="ch4", conc=dict(ppb=1800), lines=mk(2))])


CELLS = {
    # name: (config, shard) - the comment says which kernel the per-list and the merged jobs take
    "far_field": (three_molecule_cell(6000, 600, 1000, 0.001, 1013.25), None),               # W = 5000: far-field kernel, R = 4
    "direct_split": (three_molecule_cell(1500, 600, 700, 0.01, 1013.25), None),              # 10^4 points: small spans, line split
    "skewed_range": (three_molecule_cell(20000, 100, 2500, 0.001, 60.0), None),              # W = 297 on 3 x 2.4e6 points: skewed walk
    "centre_only": (three_molecule_cell(3000, 600, 700, 0.01, 1.0), None),                   # W = 1: every line adds its centre sample only
    "regrid": (three_molecule_cell(1500, 600, 700, 0.01, 10132.5, dynamic=True), None),      # work grid 0.1, np.interp onto 0.01
    "shard": (three_molecule_cell(6000, 600, 1000, 0.001, 1013.25), (4, 2)),                 # rank 2 of 4: p_begin > 0, halo lines
}


@pytest.mark.parametrize("name", list(CELLS))
def test_merged_step_equals_per_list_step(ctx, name):
    from pyrad_amd import engine
    cfg, shard = CELLS[name]
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols_of(cfg),
                             cfg["base_resolution"], cfg["dynamic_resolution"], shard=shard)
    L.enqueue(surface_T=288.0)
    ref = L.results()
    for b in (L.abs_coef, L.trans, L.I_out):
        b.fill(float("nan"))
    L.enqueue(surface_T=288.0, merged=True)
    got = L.results()
    sl = slice(L.first, L.first + L.count)
    assert np.all(np.isfinite(got["abs_coef"][sl])) and np.any(ref["abs_coef"][sl] > 0)
    # same contributions, summed in another order, the factor on the amplitude instead of on the sum
    assert rel_err(got["abs_coef"][sl], ref["abs_coef"][sl]) <= 2e-13
    k, d = ref["abs_coef"][sl], cfg["depth"]
    assert np.all(np.abs(got["transmittance"][sl] - ref["transmittance"][sl]) <= (2e-13 * k * d + 1e-15) * ref["transmittance"][sl] + 1e-300)
    assert rel_err(got["transmission"][sl], ref["transmission"][sl]) <= 1e-11
    # bit-identical reruns (no atomics, fixed order per point)
    L.enqueue(surface_T=288.0, merged=True)
    again = L.results()
    assert all(np.array_equal(again[k_][sl], got[k_][sl]) for k_ in got)
    if shard is not None:      # nothing outside the shard is touched (abs_coef was poisoned before the merged step)
        assert np.all(np.isnan(got["abs_coef"][:L.first])) and np.all(np.isnan(got["abs_coef"][L.first + L.count:L.n]))
    # the regime counters (cls:368-370, 406) of a merged batch are still per line list
    if shard is None:
        merged_counts = ctx.last_regime_counts(len(L.jobs))
        L.enqueue_xsec()
        assert np.array_equal(ctx.last_regime_counts(len(L.jobs)), merged_counts)
        assert merged_counts.sum() == L.n_lines
    # only the absorption coefficient asked for: the kernel's plain store gives the same bits
    only_k = ctx.buffer(L.padded_n).fill(float("nan"))
    ctx.layer_merged_step_dev([j[0] for j in L.jobs], [j[1] for j in L.jobs], L.grid_native, L.iso_mol, L.conc, L.depth,
                              abs_coef=only_k)
    assert np.array_equal(only_k.download(L.n)[sl], got["abs_coef"][sl])
    only_k.free()
    L.free()


def test_merged_step_with_duplicate_centres_and_empty_lists(ctx):
    """ties across lists (the same wavenumbers in two lists), a list without lines, a single list: the merged order is
    the stable merge, and a one-list job is the plain job with the factor on its amplitudes"""
    from pyrad_amd import engine
    base = synthetic.make_lines(11, 2000, 590, 710)
    empty = {k: v[:0] for k, v in base.items()}
    cfg = dict(depth=10.0, T=296, P=1013.25, range_min=600, range_max=700, base_resolution=0.01, dynamic_resolution=True,
               molecules=[dict(species="co2", conc=dict(ppm=400), lines=base), dict(species="h2o", conc={"%": 1.0}, lines=base),
                          dict(species="ch4", conc=dict(ppb=1800), lines=empty)])
    for mols in (mols_of(cfg), mols_of(cfg)[:1]):
        L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                 cfg["base_resolution"], cfg["dynamic_resolution"])
        L.enqueue(surface_T=288.0)
        ref = L.results()
        L.abs_coef.fill(float("nan"))
        L.enqueue(surface_T=288.0, merged=True)
        got = L.results()
        assert rel_err(got["abs_coef"], ref["abs_coef"]) <= 2e-13
        L.free()


def test_merged_column_equals_per_list_column(ctx):
    from pyrad_amd import engine
    col = synthetic.config_c5(n_layers=6, n_lines=4000, range_min=600, range_max=900)
    cfgs = [dict(c, molecules=mols_of(c)) for c in col["layers"]]
    column = engine.ResidentColumn(ctx, cfgs, col["surface_T"])
    column.enqueue(layer_arrays=True)
    ref = column.results()
    ref_k = [L.abs_coef.download(column.n) for L in column.layers]
    for L in column.layers:
        L.abs_coef.fill(float("nan")); L.trans.fill(float("nan"))
    column.enqueue(layer_arrays=True, merged=True)
    got = column.results()
    for L, k in zip(column.layers, ref_k):
        assert rel_err(L.abs_coef.download(column.n), k) <= 2e-13
    for a, b, k, L in zip(got["transmittance"], ref["transmittance"], ref_k, column.layers):
        assert np.all(np.abs(a - b) <= (2e-13 * k * L.depth + 1e-15) * b + 1e-300)
    assert rel_err(got["toa"], ref["toa"]) <= 1e-12
    column.enqueue(layer_arrays=False, merged=True)             # without the per-layer transmittances: same outgoing spectrum
    assert np.array_equal(column.results()["toa"], got["toa"])
    column.free()


def test_merged_entry_points_refuse_bad_arguments(ctx):
    from pyrad_amd import _native as nat, engine
    cfg, _ = CELLS["direct_split"]
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols_of(cfg),
                             cfg["base_resolution"], cfg["dynamic_resolution"])
    lines, iso = [j[0] for j in L.jobs], [j[1] for j in L.jobs]
    with pytest.raises(nat.LblError) as e:          # no output array
        ctx.layer_merged_step_dev(lines, iso, L.grid_native, L.iso_mol, L.conc, L.depth)
    assert e.value.code == -1
    with pytest.raises(nat.LblError) as e:          # iso_mol must be non-decreasing and < n_mol
        ctx.layer_merged_step_dev(lines, iso, L.grid_native, [0, 2, 1], L.conc, L.depth, abs_coef=L.abs_coef)
    assert e.value.code == -1
    other = nat.IsoParams(iso[1].T + 1.0, iso[1].P, iso[1].q_frac, iso[1].molmass, iso[1].Q_T, iso[1].Q_296)
    with pytest.raises(nat.LblError) as e:          # all lists of a layer share its T and P
        ctx.layer_merged_step_dev(lines, [iso[0], other, iso[2]], L.grid_native, L.iso_mol, L.conc, L.depth, abs_coef=L.abs_coef)
    assert e.value.code == -1
    short = ctx.buffer(10)
    with pytest.raises(nat.LblError) as e:
        ctx.layer_merged_step_dev(lines, iso, L.grid_native, L.iso_mol, L.conc, L.depth, abs_coef=short)
    assert e.value.code == -1
    ctx.set_option("accum_variant", 0)              # the scalar-cache kernel has no merged form
    try:
        with pytest.raises(nat.LblError) as e:
            ctx.layer_merged_step_dev(lines, iso, L.grid_native, L.iso_mol, L.conc, L.depth, abs_coef=L.abs_coef)
        assert e.value.code == -1
    finally:
        ctx.set_option("accum_variant", 5)
    with pytest.raises(nat.LblError) as e:          # swept range outside the grid: overflow-safe check
        ctx.column_fold_dev([L.abs_coef], [L.T], [L.depth], L.range_min, L.range_max, L.n, L.I_out, surface_T=288.0,
                            first=2**62, count=2**62)
    assert e.value.code == -1
    with pytest.raises(nat.LblError) as e:
        ctx.layer_sweep_dev([j[3] for j in L.jobs], L.iso_mol, L.conc, L.P, L.T, L.depth, L.range_min, L.range_max, L.n,
                            abs_coef=L.abs_coef, first=2**62, count=2**62)
    assert e.value.code == -1
    with pytest.raises(nat.LblError) as e:          # need I_in or surface_T
        ctx.column_fold_dev([L.abs_coef], [L.T], [L.depth], L.range_min, L.range_max, L.n, L.I_out)
    assert e.value.code == -1
    short.free()
    # the merged step still works after the refusals, and is what the per-list step gives
    L.enqueue(surface_T=288.0)
    ref = L.results()
    L.enqueue(surface_T=288.0, merged=True)
    assert rel_err(L.results()["abs_coef"], ref["abs_coef"]) <= 2e-13
    L.free()


def test_a_captured_step_goes_stale_when_an_option_changes(ctx):
    """advisor, round 4: lbl_set_option("accuracy" ...) selects other kernels and term counts; a graph captured before
    must not go on replaying the old arithmetic: the option change bumps the context's epoch, lbl_graph_launch answers
    LBL_ERR_STATE, engine.StepGraph captures again by itself."""
    from pyrad_amd import _native as nat, engine
    cfg, _ = CELLS["far_field"]
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols_of(cfg),
                             cfg["base_resolution"], cfg["dynamic_resolution"])
    for merged in (False, True):
        g = L.capture_step(surface_T=288.0, merged=merged)
        g.launch()
        exact = L.results()["abs_coef"]
        ctx.set_option("accuracy", 1)
        try:
            with pytest.raises(nat.LblError) as e:
                g.g.launch()                                 # the raw graph: stale
            assert e.value.code == -6
            g.launch()                                       # StepGraph: this step kernel by kernel, then a new capture
            assert g.recaptures == 1
            budget = L.results()["abs_coef"]
            L.enqueue(surface_T=288.0, merged=merged)
            assert np.array_equal(L.results()["abs_coef"], budget)
            assert not np.array_equal(budget, exact) and rel_err(budget, exact) <= 1e-9
            ctx.set_option("accuracy", 1)                    # same value again: nothing changed, the graph stays valid
            g.g.launch()
        finally:
            ctx.set_option("accuracy", 0)
        g.launch()
        assert np.array_equal(L.results()["abs_coef"], exact)
        g.free()
    L.free()


@pytest.mark.parametrize("seed", range(24))
def test_merged_random_cells(ctx, seed):
    """Seeded random cells through the merged step against the per-list step AND the oracle: windows from a single point
    to several thousand (every kernel route: small-span direct, skewed walk with 1 / 2 / 4 waves per span, far-field with
    the edge walk or the edge series at 12 / 15 / 20 terms), one to five molecules, dense and sparse lists, duplicate
    wavenumbers across lists, a list without lines, dynamic resolution (regrid) and both accuracy modes."""
    from oracle import pyrad_oracle as orc
    from conftest import point_tolerance, rel_err_points
    from pyrad_amd import engine
    from pyrad_amd.model import concentration_from_kwargs
    rng = np.random.default_rng(9000 + seed)
    base = float(rng.choice([0.01, 0.001]))
    P = float(rng.choice([1013.25, 1013.25, 600.0, 250.0, 90.0, 20.0, 3.0, 0.15]))
    if seed % 6 == 5:
        P = float(rng.choice([3000.0, 10132.5]))           # coarse work grid with dynamic resolution -> regrid
    dynamic = bool(seed % 6 == 5)
    T = int(rng.integers(180, 340))
    n_pts = int(rng.choice([3000, 20000, 60000, 400000])) if base == 0.001 else int(rng.choice([1500, 10000, 40000]))
    rmin = float(rng.integers(5, 2000))
    rmax = rmin + n_pts * base
    n_mol = int(rng.integers(1, 6))
    species = ["co2", "h2o", "ch4", "o3", "co2_636"][:n_mol]
    dens = float(rng.choice([0.02, 0.06, 0.2]))             # lines per grid point and list
    mols_cfg = []
    first = None
    for i, sp_name in enumerate(species):
        n_lines = max(int(n_pts * dens), 3) if not (i == 2 and seed % 4 == 0) else 0
        dfc = 5.0 * P / 1013.25
        lines = synthetic.make_lines(100 * seed + i, max(n_lines, 1), max(rmin - dfc - 1, 0.01), rmax + dfc + 1)
        if n_lines == 0:
            lines = {k: v[:0] for k, v in lines.items()}
        elif first is not None and seed % 3 == 0:           # ties across lists: the same wavenumbers as the first list's first half
            k = min(len(first["nu"]) // 2, len(lines["nu"]))
            lines["nu"] = np.sort(np.concatenate([first["nu"][:k], lines["nu"][k:]]))
        if first is None:
            first = lines
        conc = [dict(ppm=400), {"%": 1.0}, dict(ppb=1800), dict(ppm=3.0), dict(ppm=4.0)][i]
        mols_cfg.append(dict(species=sp_name, conc=conc, lines=lines))
    cfg = dict(depth=float(rng.uniform(1, 1e4)), T=T, P=P, range_min=rmin, range_max=rmax, base_resolution=base,
               dynamic_resolution=dynamic, molecules=mols_cfg)
    mols = mols_of(cfg)
    L = engine.ResidentLayer(ctx, cfg["depth"], T, P, rmin, rmax, mols, base, dynamic)
    g = L.g
    budget = seed % 5 == 4
    ctx.set_option("accuracy", 1 if budget else 0)
    try:
        L.enqueue(surface_T=288.0)
        ref = L.results()
        for b in (L.abs_coef, L.trans, L.I_out):
            b.fill(float("nan"))
        L.enqueue(surface_T=288.0, merged=True)
        got = L.results()
    finally:
        ctx.set_option("accuracy", 0)
    assert np.all(np.isfinite(got["abs_coef"]))
    tol_pair = 2e-13 if not budget else 2e-9                # (budget: the two paths cut different lines' series at 1e-9)
    assert rel_err(got["abs_coef"], ref["abs_coef"]) <= tol_pair, (seed, P, base, g["W"], n_mol)
    # the oracle on what it finishes in a second
    if sum(len(m["lines"]["nu"]) for m in mols_cfg) * max(2 * g["W"], 1) <= 4e8:
        k_ref = np.zeros(g["n_base"])
        for m, mc in zip(mols, mols_cfg):
            iso = m["isotopologues"][0]
            sel = orc.select_window(iso["lines"], g["eff_min"], g["eff_max"])
            xs, _ = orc.create_cross_section(sel, T, P, m["conc"], iso["molmass"], iso["q_T"], iso["q296"],
                                             orc.layer_grid(P, rmin, rmax, base, dynamic))
            k_ref = k_ref + orc.abs_coef(np.zeros(g["n_base"]) + xs, m["conc"], P, T)
        xa = orc.x_axis(rmin, rmax, base)
        tol = point_tolerance(xa, T, g["dfc"], rtol_base=(1e-9 if budget else 2e-12))
        e = rel_err_points(got["abs_coef"], k_ref)
        assert np.all(e <= tol), (seed, float(e.max()), P, base, g["W"], n_mol)
    L.free()
