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
