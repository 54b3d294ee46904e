"""CPU: property tests (hypothesis) of the host logic — grid scalars against the oracle, shard
arithmetic, halo selection, the sharded evaluation count, mergeArray against the oracle."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from pyrad_amd import dist, engine, model
from oracle import pyrad_oracle as orc

COMMON = dict(max_examples=150, deadline=None)


@settings(**COMMON)
@given(P=st.floats(0.01, 50000.0), rmin=st.floats(0.0, 3000.0), width=st.floats(0.01, 500.0),
       base=st.sampled_from([0.01, 0.001, 0.0001]), dyn=st.booleans())
def test_layer_grid_equals_oracle(P, rmin, width, base, dyn):
    g, o = engine.layer_grid(P, rmin, rmin + width, base, dyn), orc.layer_grid(P, rmin, rmin + width, base, dyn)
    for key in ("dfc", "eff_min", "eff_max", "resolution", "n_base", "n_work", "W"):
        assert g[key] == o[key], key


@settings(**COMMON)
@given(n=st.integers(0, 10**7), world=st.integers(1, 64))
def test_shard_bounds_tile_the_grid(n, world):
    covered = 0
    prev_end = 0
    S0 = None
    for rank in range(world):
        S, first, count = dist.shard_bounds(n, world, rank)
        S0 = S if S0 is None else S0
        assert S == S0 and first == min(rank * S, n) or count == 0
        assert count >= 0 and first + count <= n
        if count:
            assert first == prev_end
            prev_end = first + count
        covered += count
    assert covered == n and S0 * world >= n


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10**6), world=st.integers(1, 8), W=st.integers(1, 400), n_work=st.integers(1, 5000))
def test_halo_selection_keeps_every_contribution(seed, world, W, n_work):
    rng = np.random.default_rng(seed)
    res, rmin = 0.01, 600.0
    nu = np.sort(rmin + rng.uniform(-W * res, (n_work + W) * res, 200))
    lines = {"nu": nu}
    total = engine.eval_count(nu, rmin, res, W, n_work)
    parts = 0
    for rank in range(world):
        S, first, count = dist.shard_bounds(n_work, world, rank)
        if count == 0:          # an empty shard does no work (a count of 0 means "unsharded" to eval_count and the C ABI)
            continue
        sel = dist.halo_select(lines, rmin, res, W, first, count)
        # dropping the lines outside the halo must not change this shard's count
        mine = engine.eval_count(sel["nu"], rmin, res, W, n_work, (first, count))
        assert mine == engine.eval_count(nu, rmin, res, W, n_work, (first, count))
        parts += mine
    assert parts == total


@settings(max_examples=100, deadline=None)
@given(lo=st.integers(0, 300), n_new=st.integers(2, 400), off=st.integers(-450, 450), n_old=st.integers(2, 400), seed=st.integers(0, 999))
def test_merge_array_model_equals_oracle(lo, n_new, off, n_old, seed):
    newX = (lo + np.arange(n_new)) * 0.01 + 600.0
    oldX = (lo + off + np.arange(n_old)) * 0.01 + 600.0
    oldY = np.random.default_rng(seed).random(n_old) + 0.5
    outcome = []
    for f in (model.mergeArray, orc.merge_array):
        try:
            outcome.append(("ok", np.asarray(f(newX, oldX, oldY), dtype=np.float64)))
        except (ValueError, IndexError) as e:
            outcome.append((type(e).__name__, None))
    assert outcome[0][0] == outcome[1][0]
    if outcome[0][0] == "ok":
        assert np.array_equal(outcome[0][1], outcome[1][1])


@settings(max_examples=80, deadline=None)
@given(seed=st.integers(0, 10**6), n=st.integers(0, 300), lo=st.floats(500.0, 700.0), width=st.floats(0.0, 120.0))
def test_select_window_and_concentration_equal_oracle(seed, n, lo, width):
    rng = np.random.default_rng(seed)
    lines = {f: np.sort(rng.uniform(480.0, 840.0, n)) if f == "nu" else rng.random(n) for f in
             ("nu", "sw", "a", "elower", "gamma_air", "gamma_self", "delta_air", "n_air")}
    if n:                      # exact boundary hits: the selection is strict on both sides (ut:437-438)
        lines["nu"][0] = lo
        lines["nu"][-1] = max(lo + width, lines["nu"][-1])
        lines["nu"].sort()
    a, b = engine.select_window(lines, lo, lo + width), orc.select_window(lines, lo, lo + width)
    for f in lines:
        assert np.array_equal(a[f], b[f]), f
    v = float(rng.uniform(0.0, 500.0))
    for key in ("ppm", "ppb", "percentage", "perc", "%", "concentration"):
        assert model.concentration_from_kwargs(**{key: v}) == orc.concentration(**{key: v}), key


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10**6), n=st.integers(0, 120), lo=st.integers(5, 30), span=st.integers(1, 4))
def test_data_dir_round_trip_equals_memory_source(tmp_path_factory, seed, n, lo, span):
    """A PyRad data/ tree written from a line list and read back gives what the in-memory source gives
    for the same window (strict bounds, last duplicate wins, 100 cm^-1 segment files)."""
    from pyrad_amd import data, synthetic
    rng = np.random.default_rng(seed)
    rmin, rmax = 100.0 * lo + 37.5, 100.0 * (lo + span) - 12.25
    nu = np.round(np.sort(rng.uniform(rmin - 60.0, rmax + 60.0, n)), 6)
    if n > 3:
        nu[1] = nu[0]                       # a duplicated wavenumber: the later row wins (ut:447)
    lines = {f: nu if f == "nu" else rng.random(n) for f in synthetic.FIELDS}
    root = str(tmp_path_factory.mktemp("pyr"))
    q = {T: 100.0 + T for T in (250, 296)}
    params = synthetic.mol_params("co2")
    data.PyradDataDir.write_tree(root, 7, lines, q, params)
    if n:
        for seg in data.PyradDataDir.segments(rmin, rmax):        # empty segments exist as empty files in PyRad
            path = "%s/7/%s.pyr" % (root, seg)
            import os
            if not os.path.isfile(path):
                open(path, "w").close()
    disk = data.PyradDataDir(root)
    mem = data.MemorySource()
    mem.register(7, lines, q, params)
    if n == 0:
        return
    a, b = disk.gatherData(7, rmin, rmax), mem.gatherData(7, rmin, rmax)
    for f in synthetic.FIELDS:
        assert np.array_equal(a[f], b[f]), f
    assert disk.getQData(7) == q and disk.readMolParams(7)[0] == params[0]


@settings(max_examples=120, deadline=None)
@given(n=st.integers(1, 3_000_000), world=st.integers(1, 16), n_lines=st.integers(0, 3000), H=st.integers(0, 6000),
       seed=st.integers(0, 2**31 - 1), clustered=st.booleans(), ratio=st.sampled_from([1.0, 1.125, 2.0, 8.0]))
def test_balanced_plan_properties(n, world, n_lines, H, seed, clustered, ratio):
    """Cost-balanced shard plans: every rank derives the same plan; shards tile the grid contiguously; boundaries
    are multiples of a workgroup's points; no shard exceeds the cap (so the all-gather slot is bounded); the
    padded gather layout round-trips; and with a loose cap the summed model cost per shard is never worse than
    one aligned block above the equal-width plan's worst shard."""
    rng = np.random.default_rng(seed)
    if clustered:
        c = np.sort((rng.normal(0.3 * n, 0.05 * n + 1, n_lines)).astype(np.int64))
    else:
        c = np.sort(rng.integers(-H, n + H + 1, n_lines))
    cost = dist.span_costs(c, H, n, has_gaussian=rng.random(n_lines) < 0.5)
    assert cost.shape == (-(-n // dist.SPAN),) and np.all(cost >= 600.0)
    plans = [dist.balanced_plan(n, world, r, cost, max_ratio=ratio) for r in range(world)]
    b = plans[0].bounds
    assert all(p.bounds == b for p in plans)
    assert b[0][0] == 0 and sum(k for _, k in b) == n
    for (f, k), (f2, _) in zip(b, b[1:] + [(n, 0)]):
        assert k >= 0 and f + k == f2 and (f % dist.ALIGN == 0 or f == n)
    n_blocks = -(-n // dist.ALIGN)
    cap_blocks = max(int(np.ceil(ratio * n_blocks / world)), 1)
    assert plans[0].S <= cap_blocks * dist.ALIGN
    spec = rng.random(n)
    gathered = np.zeros(world * plans[0].S)
    for r, (f, k) in enumerate(b):
        gathered[r * plans[0].S:r * plans[0].S + k] = spec[f:f + k]
    assert np.array_equal(plans[0].assemble(gathered), spec)
    if plans[0].in_place:                                   # then slot r starts where shard r starts
        assert all(f == min(r * plans[0].S, n) for r, (f, _) in enumerate(b))
    if ratio >= 8.0 and world > 1:
        prefix = np.concatenate([[0.0], np.cumsum(cost)])
        load = lambda bounds: max(prefix[-(-(f + k) // dist.SPAN)] - prefix[f // dist.SPAN] for f, k in bounds)
        block = float(np.max(np.add.reduceat(cost, np.arange(0, cost.size, dist.ALIGN // dist.SPAN))))
        assert load(b) <= load(dist.equal_plan(n, world, 0).bounds) + 2 * block


@settings(max_examples=40, deadline=None)
@given(world=st.integers(2, 8), seed=st.integers(0, 1000), mode=st.sampled_from(["auto", "balanced", "equal"]))
def test_choose_shards_is_rank_independent(world, seed, mode):
    """engine.choose_shards: every rank makes the same choice and gets the same bounds (pure host arithmetic
    on the line positions), whatever the mode."""
    from pyrad_amd import synthetic
    g = engine.layer_grid(1013.25, 600, 640, .001, False)
    lines = synthetic.make_lines(seed, 400, g["eff_min"], g["eff_max"])
    lines["nu"] = np.sort(600.0 + (lines["nu"] - lines["nu"].min()) * (0.25 + (seed % 4) * 0.25))   # more or less clustered
    cfgs = [dict(depth=1.0, T=280, P=1013.25, range_min=600, range_max=640, base_resolution=.001, dynamic_resolution=False,
                 molecules=[dict(conc=4e-4, isotopologues=[dict(lines=lines, molmass=43.98983, q_T=1.0, q296=1.0)])])]
    got = [engine.choose_shards(cfgs, world, r, mode) for r in range(world)]
    assert all(p.bounds == got[0][0].bounds and what == got[0][1] for p, what in got)
    assert [p.rank for p, _ in got] == list(range(world))
    assert sum(k for _, k in got[0][0].bounds) == g["n_work"]
    if mode == "equal":
        assert got[0][0].in_place and got[0][1] == "equal"
