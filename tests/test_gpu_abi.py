"""GPU: hardening of the C boundary - no C++ exception crosses it, absurd sizes come back as
status codes, and the padded-gather compaction validates its arguments."""
import numpy as np
import pytest

from pyrad_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pyrad_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def test_exceptions_become_status_codes(ctx):
    from pyrad_amd import _native as nat
    for value, code in ((1, -5), (2, -6), (3, -5)):      # bad_alloc -> OOM, runtime_error -> STATE, length_error -> OOM
        with pytest.raises(nat.LblError) as e:
            ctx.set_option("debug_throw", value)
        assert e.value.code == code, (value, str(e.value))
    ctx.set_option("debug_throw", 0)                      # and the context is still usable
    assert "gfx950" in ctx.device_info()["name"]


def test_absurd_job_count_is_rejected_not_allocated(ctx):
    from pyrad_amd import _native as nat, engine
    C = nat.C
    g = engine.layer_grid(1013.25, 600, 700, .01, True)
    L = ctx.lines(synthetic.make_lines(1, 10, 595, 705))
    out = ctx.buffer(g["n_base"])
    one = (nat._P * 1)(L.h)
    iso = (nat.IsoParams * 1)(nat.IsoParams(296.0, 1013.25, 4e-4, 44.0, 286.0, 286.0))
    grid = (nat.Grid * 1)(engine.native_grid(g))
    outs = (nat._P * 1)(out.h)
    rc = ctx.lib.lbl_xsec_accumulate_dev(ctx.h, 2**31 - 1, one, iso, grid, outs)
    assert rc == -1 and b"jobs per batch" in ctx.lib.lbl_last_error(ctx.h)
    rc = ctx.lib.lbl_xsec_accumulate_dev(ctx.h, -3, one, iso, grid, outs)
    assert rc == -1
    counts = (C.c_int64 * 3)()
    assert ctx.lib.lbl_last_regime_counts(ctx.h, 2**31 - 1, counts) == -1
    L.free(); out.free()


def test_gather_compact(ctx):
    from pyrad_amd import _native as nat
    rng = np.random.default_rng(3)
    bounds = [(0, 1024), (1024, 3072), (4096, 2048), (6144, 0), (6144, 856)]
    S, n = 3072, 7000
    spec = rng.random(n)
    padded = np.full(len(bounds) * S, -1.0)
    for r, (f, c) in enumerate(bounds):
        padded[r * S:r * S + c] = spec[f:f + c]
    g = ctx.buffer(len(bounds) * S).upload(padded)
    out = ctx.buffer(n).fill(0.0)
    ctx.gather_compact_dev(g, S, bounds, out)
    assert np.array_equal(out.download(n), spec)
    with pytest.raises(nat.LblError):                     # a shard longer than its slot
        ctx.gather_compact_dev(g, 1000, bounds, out)
    with pytest.raises(nat.LblError):                     # output too short
        ctx.gather_compact_dev(g, S, bounds, ctx.buffer(n - 1))
    g.free(); out.free()
