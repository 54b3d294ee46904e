"""GPU: the dispatch schedule of a launch group built on the device (span tables from the centre indices K1 wrote,
tile costs, longest-first / XCD-partitioned / bin-packed order: lbl_kernels.hip "Schedule of a launch group") against
the host build it replaces on the time-to-first-spectrum path (re-windowing after changePressure / changeRange,
pyradClasses.py:734-752 -> cls:45-56).  The span tables must be the same integers, the dispatch lists permutations of
the same (job, tile) set, and every spectrum bit-identical (outputs are pre-filled with NaN: a tile nobody ran shows)."""
import numpy as np
import pytest

from pyrad_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    from pyrad_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def molecules(cfg):
    import bench
    return bench.molecules_of(cfg)


def run_layer(ctx, L, build, full=True):
    ctx.set_option("schedule_build", build)
    for j in L.jobs:
        j[3].fill(float("nan"))
    L.enqueue(surface_T=288.0)
    ctx.sync()
    # (the absorption coefficient sums every cross section: a NaN left anywhere shows in it)
    out = ([L.xsec_host(i) for i in range(len(L.jobs))] if full else []) + [L.abs_coef.download(L.n)]
    out = [a[L.first:L.first + L.count] for a in out]            # (a shard computes its own range of the grid only)
    sched = ctx.schedule_export(0)
    return out, sched


def compare(dev, host, n_jobs_expected=None):
    (out_d, (list_d, tabs_d, on_dev)), (out_h, (list_h, tabs_h, on_dev_h)) = dev, host
    assert on_dev and not on_dev_h
    assert tabs_d.shape == tabs_h.shape and np.array_equal(tabs_d, tabs_h)
    # (a single-round launch packed per XCD has idle positions: (0, -1), workgroups that exit at once)
    list_d = list_d[list_d[:, 1] >= 0]
    assert list_d.shape == list_h.shape
    key = lambda a: np.sort(a[:, 0].astype(np.int64) * (1 << 32) + a[:, 1].astype(np.int64))
    assert np.array_equal(key(list_d), key(list_h))                      # the same workgroups, each exactly once
    assert len(np.unique(key(list_d))) == len(list_d)
    for a, b in zip(out_d, out_h):
        assert np.array_equal(a, b, equal_nan=False), "spectra differ between device- and host-built schedules"
        assert np.all(np.isfinite(a))


def test_single_round_cell_bin_packed(ctx):
    """C2's shape (one line list, 391 workgroups: one round, the bin-packed order)"""
    from pyrad_amd import engine
    cfg = synthetic.config_c2(n_lines=20000, range_min=500, range_max=900, seed=2)
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], molecules(cfg),
                             cfg["base_resolution"], False)
    dev = run_layer(ctx, L, 1)
    host = run_layer(ctx, L, 0)
    compare(dev, host)
    assert len(dev[1][0]) <= 2048
    # the order of a single-round launch: workgroup p runs on XCD p % 8 and every XCD packs its own tiles into its own CUs.
    # Below 3 workgroups per CU (this cell) an XCD's tiles are every 8th of the longest-first order; forced local
    # (`accum_xcd_pack` 2) they are contiguous runs of the sequence, 1 per XCD or `accum_xcd_chunks`.  Idle positions: (0, -1).
    def runs_per_xcd(lst):
        assert len(lst) % 8 == 0
        worst = 0
        for x in range(8):
            mine = lst[x::8]
            tiles = np.sort(mine[mine[:, 1] >= 0][:, 1])
            worst = max(worst, 1 + int(np.count_nonzero(np.diff(tiles) != 1)) if len(tiles) else 0)
        return worst
    assert runs_per_xcd(dev[1][0]) > 8
    ctx.set_option("accum_xcd_pack", 2)
    ctx.set_option("accum_xcd_tolerance", -1)
    one = run_layer(ctx, L, 1)
    assert runs_per_xcd(one[1][0]) == 1
    compare(one, host)
    ctx.set_option("accum_xcd_chunks", 4)
    four = run_layer(ctx, L, 1)
    ctx.set_option("accum_xcd_chunks", 0)
    assert 1 < runs_per_xcd(four[1][0]) <= 4
    compare(four, host)
    ctx.set_option("accum_xcd_tolerance", 0)           # (whichever order the tolerance leaves this cell: a permutation again)
    compare(run_layer(ctx, L, 1), host)
    ctx.set_option("accum_xcd_tolerance", 3)
    ctx.set_option("accum_xcd_pack", 0)                # round 4's order: one packing over all CUs, no idle positions
    old = run_layer(ctx, L, 1)
    ctx.set_option("accum_xcd_pack", 1)
    assert np.all(old[1][0][:, 1] >= 0)
    compare(old, host)
    L.free()


@pytest.mark.parametrize("shard", [None, (8, 3), (2, 1)])
def test_multi_round_cell_xcd_partitioned(ctx, shard):
    """three line lists on 100-900 cm^-1 at 0.001 (2,346 workgroups: several rounds, XCD-partitioned order); shards of it
    (a shard of 8 is one round again)"""
    from pyrad_amd import engine
    cfg = synthetic.config_c3(n_lines=30000, range_min=100, range_max=900)
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], molecules(cfg),
                             cfg["base_resolution"], False, shard=shard)
    dev = run_layer(ctx, L, 1)
    host = run_layer(ctx, L, 0)
    compare(dev, host)
    L.free()


def test_lopsided_costs_take_the_global_sort(ctx):
    """a few very dense tiles among many empty ones: one part of the XCD partition then holds most of the cheap tiles and
    does not fit the LDS sort (global-memory fallback of sched_order_xcd_kernel)"""
    from pyrad_amd import engine
    rng = np.random.default_rng(5)
    base = synthetic.make_lines(77, 60000, 1000.0, 1002.0)                 # all lines in 2 cm^-1 of a 2400 cm^-1 grid
    mol = dict(conc=4e-4, isotopologues=[dict(lines=base, molmass=synthetic.SPECIES["co2"]["molmass"],
                                              q_T=synthetic.q_value("co2", 296), q296=synthetic.SPECIES["co2"]["q296"])])
    Ls = [engine.ResidentLayer(ctx, 10.0, 296, 60.0, 100.0, 2500.0, [mol, mol, mol, mol, mol, mol, mol, mol], .0001, False)]
    L = Ls[0]                                                              # 8 x 24e6 points / 1024 per workgroup
    dev = run_layer(ctx, L, 1, full=False)
    host = run_layer(ctx, L, 0, full=False)
    compare(dev, host)
    assert len(dev[1][0]) > 8 * 16384
    L.free()


def test_single_round_runs_that_do_not_fit_fall_back_to_the_mix(ctx):
    """one round of workgroups whose cost sits in a few tiles (all lines within half a wavenumber of a 100 cm^-1 grid): a
    contiguous run worth an eighth of the cost would hold half of the tiles - more than the positions reserved per XCD - so
    the packing kernel falls back to the longest-first mix by itself; forced either way the result is the host's"""
    from pyrad_amd import engine
    base = synthetic.make_lines(78, 4000, 650.0, 650.5)
    mol = dict(conc=4e-4, isotopologues=[dict(lines=base, molmass=synthetic.SPECIES["co2"]["molmass"],
                                              q_T=synthetic.q_value("co2", 296), q296=synthetic.SPECIES["co2"]["q296"])])
    L = engine.ResidentLayer(ctx, 10.0, 296, 1013.25, 600.0, 700.0, [mol], .001, False)
    ctx.set_option("accum_line_split", 1)
    host = run_layer(ctx, L, 0)
    for pack, tol in ((2, -1), (2, 3), (3, 3), (1, 3), (0, 3)):
        ctx.set_option("accum_xcd_pack", pack)
        ctx.set_option("accum_xcd_tolerance", tol)
        dev = run_layer(ctx, L, 1)
        compare(dev, host)
        lst = dev[1][0]
        real = lst[lst[:, 1] >= 0]
        assert len(real) == len(host[1][0])
        if pack == 2:
            # fell back: no XCD holds a contiguous run of the sequence
            mine = np.sort(lst[0::8][lst[0::8][:, 1] >= 0][:, 1])
            assert len(mine) > 2 and np.count_nonzero(np.diff(mine) != 1) > 0
    ctx.set_option("accum_xcd_pack", 1)
    ctx.set_option("accum_xcd_tolerance", 3)
    ctx.set_option("accum_line_split", 0)
    L.free()


def test_column_groups(ctx):
    """a 6-layer column, 1013 -> 20 mbar: the wide layers' group (far-field kernel) and the narrow layers' group
    (skewed-range kernel) get a device-built schedule each; every layer's arrays and the outgoing spectrum agree"""
    from pyrad_amd import engine
    import bench
    cfg = synthetic.config_c5(n_layers=6, n_lines=20000, range_min=100, range_max=900)
    cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
    col = engine.ResidentColumn(ctx, cfgs, cfg["surface_T"])
    res = {}
    for build in (1, 0):
        ctx.set_option("schedule_build", build)
        for j in col.jobs:
            j[3].fill(float("nan"))
        col.enqueue(layer_arrays=True)
        ctx.sync()
        r = col.results()
        res[build] = ([j[3].download(col.n) for j in col.jobs] + [r["toa"]] + r["transmittance"],
                      [ctx.schedule_export(k) for k in range(2)])
    for k in range(2):
        compare((res[1][0], res[1][1][k]), (res[0][0], res[0][1][k]))
    col.free()


def test_schedule_is_built_once_and_reused(ctx):
    from pyrad_amd import engine
    cfg = synthetic.config_c2(n_lines=5000, range_min=600, range_max=700, seed=4)
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], molecules(cfg),
                             cfg["base_resolution"], False)
    a, s0 = run_layer(ctx, L, 1)
    b, s1 = run_layer(ctx, L, 1)
    assert np.array_equal(s0[0], s1[0]) and np.array_equal(s0[1], s1[1])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    g = L.capture_step(surface_T=288.0)           # the built schedule can be captured; nothing pends
    g.launch()
    ctx.sync()
    g.free()
    L.free()


@pytest.mark.parametrize("seed", range(24))
def test_random_cells_device_schedule_equals_host_schedule_and_in_kernel_search(ctx, seed):
    """random gas cells (windows from 1 point to thousands, regrid on and off, empty and tiny line lists, grids from a few
    points to 60,000, so every launch shape: line splits, small spans, the skewed-range routing when forced) through the
    resident path: device-built schedule, host-built schedule and no schedule at all give the same bits"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_parity import random_cell, resident_xsec
    from oracle import pyrad_oracle as orc
    rng = np.random.default_rng(4000 + seed)
    cell = random_cell(rng, seed, orc)
    if cell is None:
        pytest.skip("degenerate grid (the reference raises)")
    lines, species, conc, T, P, rmin, rmax, base, dyn, g = cell
    from pyrad_amd import engine
    L = ctx.lines(engine.select_window(lines, g["eff_min"], g["eff_max"]))
    got = {}
    try:
        for skew in (1, 2):
            ctx.set_option("accum_skew", skew)
            for tag, lpt, build in (("search", 0, 1), ("device", 4, 1), ("host", 4, 0)):
                ctx.set_option("accum_longest_first", lpt)
                ctx.set_option("schedule_build", build)
                got[tag] = resident_xsec(ctx, L, species, conc, T, P, rmin, rmax, base, dyn)
            for tag in ("device", "host"):
                assert got[tag][1] == got["search"][1]
                assert np.array_equal(got[tag][0], got["search"][0], equal_nan=True), (tag, skew, g["W"], len(lines["nu"]))
            assert np.all(np.isfinite(got["device"][0]))
    finally:
        ctx.set_option("accum_skew", 1)
        ctx.set_option("accum_longest_first", 4)
        ctx.set_option("schedule_build", 1)
        L.free()
