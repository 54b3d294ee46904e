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


def test_graph_capture_replays_the_step_bit_for_bit(ctx):
    """lbl_capture_begin / _end / lbl_graph_launch: a layer step (three line lists: line prep, accumulate,
    sweep) and a column step replayed from their graphs give the arrays of the kernel-by-kernel route;
    an unprimed sequence cannot be captured (and the context survives); a graph whose descriptor slot
    was taken by other batches refuses to launch."""
    from pyrad_amd import _native as nat, engine
    g = engine.layer_grid(1013.25, 640, 660, .001, False)
    mols = []
    for s_, seed, conc in (("co2", 61, 4e-4), ("h2o", 62, 1e-2), ("ch4", 63, 1.8e-6)):
        sp = synthetic.SPECIES[s_]
        mols.append(dict(conc=conc, isotopologues=[dict(lines=synthetic.make_lines(seed, 700, g["eff_min"], g["eff_max"]),
                                                        molmass=sp["molmass"], q_T=synthetic.q_value(s_, 270), q296=sp["q296"])]))
    L = engine.ResidentLayer(ctx, 10.0, 270, 1013.25, 640, 660, mols, .001, False)
    with pytest.raises(nat.LblError) as e:                # nothing has run yet: scratch, schedule, descriptors are missing
        ctx.capture(lambda: L.enqueue(surface_T=288.0))
    assert e.value.code == -6 and "capture" in str(e.value)
    L.enqueue(surface_T=288.0)                            # the context is still usable
    ref = L.results()
    graph = L.capture_step(surface_T=288.0)
    for b in (L.abs_coef, L.trans, L.I_out):
        b.fill(0.0)
    graph.launch()
    got = L.results()
    assert all(np.array_equal(got[k], ref[k]) for k in ref) and np.all(ref["abs_coef"] > 0)
    graph.launch(); graph.launch()
    assert all(np.array_equal(L.results()[k], ref[k]) for k in ref)
    # a single-line-list layer (sweep fused into the accumulate kernel) and a column
    one = engine.ResidentLayer(ctx, 10.0, 270, 1013.25, 640, 660, mols[:1], .001, False)
    one.enqueue(surface_T=288.0)
    ref1 = one.results()
    g1 = one.capture_step(surface_T=288.0)
    one.I_out.fill(0.0)
    g1.launch()
    assert np.array_equal(one.results()["transmission"], ref1["transmission"])
    cfgs = [dict(depth=1e4, T=T, P=P, range_min=640, range_max=660, base_resolution=.001, dynamic_resolution=False, molecules=mols)
            for T, P in ((285, 1013.25), (250, 300.0), (220, 30.0))]
    col = engine.ResidentColumn(ctx, cfgs, 288.0)
    col.enqueue(layer_arrays=False)
    toa = col.results()["toa"]
    gc = col.capture_step(layer_arrays=False)
    col.I_toa.fill(0.0)
    gc.launch()
    assert np.array_equal(col.results()["toa"], toa)
    # five other batches take the four descriptor slots: the graphs of the first layer are stale now
    others = []
    for k in range(5):
        o = engine.ResidentLayer(ctx, 10.0, 260 + k, 1013.25, 640, 660, mols[:2], .001, False)
        o.enqueue(surface_T=288.0)
        others.append(o)
    with pytest.raises(nat.LblError) as e:
        graph.g.launch()                                  # (the library's graph object: lbl_graph_launch refuses)
    assert e.value.code == -6 and "stale" in str(e.value)
    L.enqueue(surface_T=288.0)                            # the kernel-by-kernel route still works, and a new capture too
    g2 = L.capture_step(surface_T=288.0)
    g2.launch()
    assert all(np.array_equal(L.results()[k], ref[k]) for k in ref)
    # freeing ANY buffer or line list of the context makes its graphs stale too: a captured kernel node may hold the address
    ctx.buffer(16).free()
    with pytest.raises(nat.LblError) as e:
        g2.g.launch()
    assert e.value.code == -6 and "stale" in str(e.value)
    # engine.StepGraph (what capture_step returns) takes a stale graph in its stride: the step runs kernel by kernel
    # once and is recorded again for the launches that follow
    for b in (L.abs_coef, L.trans, L.I_out):
        b.fill(0.0)
    g2.launch()
    assert g2.recaptures == 1 and all(np.array_equal(L.results()[k], ref[k]) for k in ref)
    L.abs_coef.fill(0.0)
    g2.launch()
    assert g2.recaptures == 1 and np.array_equal(L.results()["abs_coef"], ref["abs_coef"])
    for x in (graph, g1, gc, g2):
        x.free()
    for o in others + [L, one, col]:
        o.free()


def test_empty_shard_does_nothing_and_counts_nothing(ctx):
    """More ranks than aligned blocks: a rank whose shard is empty must not fall back to the whole grid
    (shard_count == 0 means 'whole grid' at the C boundary): it enqueues nothing and reports 0 evals."""
    from pyrad_amd import engine, dist
    g = engine.layer_grid(1013.25, 600, 603, .001, False)
    sp = synthetic.SPECIES["co2"]
    mols = [dict(conc=4e-4, isotopologues=[dict(lines=synthetic.make_lines(5, 200, g["eff_min"], g["eff_max"]),
                                                molmass=sp["molmass"], q_T=286.09, q296=sp["q296"])])]
    n = g["n_work"]
    plans = [engine.balanced_shards([dict(depth=1.0, T=296, P=1013.25, range_min=600, range_max=603, base_resolution=.001,
                                          dynamic_resolution=False, molecules=mols)], 8, r) for r in range(8)]
    empties = [r for r in range(8) if plans[r].count == 0]
    assert empties and sum(p.count for p in plans[:1]) >= 0 and sum(c for _, c in plans[0].bounds) == n
    whole = engine.ResidentLayer(ctx, 1.0, 296, 1013.25, 600, 603, mols, .001, False)
    total = 0
    for r in range(8):
        L = engine.ResidentLayer(ctx, 1.0, 296, 1013.25, 600, 603, mols, .001, False, shard=plans[r])
        L.enqueue(surface_T=288.0)
        ctx.sync()
        if r in empties:
            assert L.empty and L.evals == 0 and not L.abs_coef.download(n).any()
        total += L.evals
        L.free()
    assert total == whole.evals
    whole.free()


def test_sweep_divisions_are_ieee_exact(ctx):
    """The sweeps divide by launch-uniform constants (1E4, k, T) with a 5-instruction sequence instead of
    the general divide; it must return the IEEE quotient bit for bit, as NumPy's crossSection *
    concentration * P / 1E4 / k / T does (pyradClasses.py:583): random cross sections over 25 decades,
    zeros, integer and non-integer temperatures, three molecules with one or two isotopologues; one pass over
    520 decades (1e-270 .. 1e+250)."""
    k_B = 1.38064852E-23
    rng = np.random.default_rng(77)
    n = 200_000
    # (round 4: the sweeps' default arithmetic is cross section x one host-computed factor - a few 1e-16 from this chain,
    # checked below; "sweep_ieee_divisions" 1 selects the reference's own rounding chain, which is what must be bit-exact)
    ctx.set_option("sweep_ieee_divisions", 1)
    cases = [(296, 1013.25, -42, -16), (217, 10.0, -42, -16), (287.65, 843.21, -42, -16), (1.0, 1e-3, -42, -16),
             (3000, 2e5, -42, -16), (255.99999999999997, 500.0, -42, -16),
             (296, 1013.25, -270, 250)]          # the whole range the 5-instruction form is stated for (lbl_kernels.hip div_uniform)
    for T, P, lo, hi in cases:
        xs = [10.0 ** rng.uniform(lo, hi, n) for _ in range(4)]
        for a in xs:
            a[rng.integers(0, n, 500)] = 0.0
        conc = [float(c) for c in (4e-4, 0.0123456789, 1.8e-6)]
        iso_mol = [0, 0, 1, 2]
        bufs = [ctx.buffer(n).upload(a) for a in xs]
        k_dev, t_dev = ctx.buffer(n), ctx.buffer(n)
        ctx.layer_sweep_dev(bufs, iso_mol, conc, P, T, 12.5, 600.0, 800.0, n, abs_coef=k_dev, trans=t_dev)
        k_ref = np.zeros(n)
        for m in range(3):
            xs_m = np.zeros(n)
            for i, a in enumerate(xs):
                if iso_mol[i] == m:
                    xs_m = xs_m + a
            k_ref = k_ref + xs_m * conc[m] * P / 1E4 / k_B / T
        assert np.array_equal(k_dev.download(n), k_ref), (T, P)
        # the column step and the single-line-list fused step use the same helper
        out = ctx.buffer(n)
        k_col = ctx.buffer(n)
        ctx.column_step_dev([dict(xsec=bufs, iso_mol=iso_mol, conc=conc, P=P, T=T, depth=12.5, abs_coef=k_col, trans=None)],
                            600.0, 800.0, n, out, surface_T=288.0)
        assert np.array_equal(k_col.download(n), k_ref), (T, P)
        # the default arithmetic (one factor from the host): a few 1e-16 from the chain, zeros exactly zero
        ctx.set_option("sweep_ieee_divisions", 0)
        ctx.layer_sweep_dev(bufs, iso_mol, conc, P, T, 12.5, 600.0, 800.0, n, abs_coef=k_dev, trans=t_dev)
        k_fast = k_dev.download(n)
        ctx.set_option("sweep_ieee_divisions", 1)
        ok = k_ref > 1e-290                                   # (below: the chain's own intermediate products go subnormal)
        assert np.all(np.abs(k_fast[ok] - k_ref[ok]) <= 1e-15 * k_ref[ok]) and np.all(k_fast[k_ref == 0] == 0)
        for b in bufs + [k_dev, t_dev, out, k_col]:
            b.free()
    ctx.set_option("sweep_ieee_divisions", 0)


def test_plain_c_host_reproduces_the_reference_peak(tmp_path):
    """examples/abi_smoke.c (gcc -std=c99, no Python, no C++): one CO2-like line through lbl_xsec_accumulate;
    the peak equals the reference's measured value 4.5462648814858876e-20 to 1e-12, the line
    is counted in the pseudo-Voigt regime and its support is [4502..5498]."""
    import os, subprocess
    from conftest import REPO
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(REPO, "pyrad_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "abi_smoke.c"),
                           "-L", lib_dir, "-lpyrad_hip", "-Wl,-rpath," + lib_dir, "-lm", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0/0/1" in p.stdout and "[4502..5498]" in p.stdout


def test_chained_contexts_give_the_same_arrays():
    """lbl_ctx_chain_accumulate: two contexts as a software pipeline (the accumulate kernels of one wait for the
    other's; line prep and sweeps do not).  Steps dealt alternately to the two contexts reproduce the arrays of a
    lone context bit for bit, with and without the chaining, and unchaining works."""
    from pyrad_amd import _native as nat, engine
    g = engine.layer_grid(1013.25, 640, 700, .001, False)
    mols = []
    for s_, seed, conc in (("co2", 71, 4e-4), ("h2o", 72, 1e-2)):
        sp = synthetic.SPECIES[s_]
        mols.append(dict(conc=conc, isotopologues=[dict(lines=synthetic.make_lines(seed, 3000, g["eff_min"], g["eff_max"]),
                                                        molmass=sp["molmass"], q_T=synthetic.q_value(s_, 280), q296=sp["q296"])]))
    a, b = nat.Context(0), nat.Context(0)
    try:
        La, Lb = (engine.ResidentLayer(c, 10.0, 280, 1013.25, 640, 700, mols, .001, False) for c in (a, b))
        La.enqueue(surface_T=288.0)
        ref = La.results()
        a.chain_accumulate(b)
        b.chain_accumulate(a)
        for k in range(6):
            (La, Lb)[k % 2].enqueue(surface_T=288.0)
        ra, rb = La.results(), Lb.results()
        assert all(np.array_equal(ra[k], ref[k]) and np.array_equal(rb[k], ref[k]) for k in ref)
        with pytest.raises(nat.LblError):
            a.chain_accumulate(a)
        a.chain_accumulate(None)
        b.chain_accumulate(None)
        La.enqueue(surface_T=288.0)
        assert all(np.array_equal(La.results()[k], ref[k]) for k in ref)
        La.free(); Lb.free()
    finally:
        a.close(); b.close()


def test_line_list_views_share_the_parent_arrays(ctx):
    """lbl_lines_view: a wavenumber window of a resident list as a view (no copy) gives bit for bit the cross section
    of the same lines uploaded on their own; a view of a view works; a list with live views cannot be destroyed."""
    from pyrad_amd import _native as nat, engine
    g = engine.layer_grid(300.0, 640, 700, .001, False)
    base = synthetic.make_lines(81, 6000, 600.0, 740.0)
    sp = synthetic.SPECIES["co2"]
    iso = nat.IsoParams(250.0, 300.0, 4e-4, sp["molmass"], synthetic.q_value("co2", 250), sp["q296"])
    grid = engine.native_grid(g)
    nu = base["nu"]
    first, end = int(np.searchsorted(nu, g["eff_min"], "right")), int(np.searchsorted(nu, g["eff_max"], "left"))
    sub = {k: v[first:end] for k, v in base.items()}
    own = ctx.lines(sub)
    master = ctx.lines(base)
    view = master.view(first, end - first)
    inner = master.view(first - 10, end - first + 30).view(10, end - first)
    outs = [ctx.buffer(g["n_base"]) for _ in range(3)]
    ctx.xsec_accumulate_dev([(own, iso, grid, outs[0]), (view, iso, grid, outs[1]), (inner, iso, grid, outs[2])])
    a, b, c = (o.download(g["n_base"]) for o in outs)
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.all(a > 0)
    with pytest.raises(nat.LblError) as e:
        master.free()
    assert e.value.code == -6 and "views" in str(e.value)
    with pytest.raises(nat.LblError):
        master.view(5990, 20)                          # outside the list
    for x in (view, inner, own):
        x.free()
    intermediate = [c_ for c_ in ctx._children if isinstance(c_, nat.Lines) and c_ is not master]
    for x in intermediate:                                # the intermediate view of `inner`
        x.free()
    master.free()
    for o in outs:
        o.free()


def test_closing_a_context_under_a_live_column_frees_views_before_lists():
    """Context.close() with a ResidentColumn (line-list masters + one view per layer and molecule) still alive: views go
    before the lists they window, nothing raises, nothing leaks (lbl_ctx_destroy refuses a context with live objects)."""
    from pyrad_amd import _native as nat, engine
    g = engine.layer_grid(1013.25, 640, 660, .001, False)
    lines = synthetic.make_lines(91, 2000, 630.0, 670.0)
    sp = synthetic.SPECIES["co2"]
    mols = [dict(conc=4e-4, isotopologues=[dict(lines=lines, molmass=sp["molmass"], q_T=synthetic.q_value("co2", 270), q296=sp["q296"])])]
    c = nat.Context(0)
    cfgs = [dict(depth=1e4, T=270, P=P, range_min=640, range_max=660, base_resolution=.001, dynamic_resolution=False, molecules=mols)
            for P in (1013.25, 300.0, 30.0)]
    col = engine.ResidentColumn(c, cfgs, 288.0)
    col.enqueue(layer_arrays=False)
    assert any(isinstance(x, nat.Lines) and x._parent is not None for x in c._children)      # views exist
    c.close()                                             # must not raise "views still alive"
    assert c.h is None


def test_destroying_a_chained_predecessor_unlinks_it():
    """lbl_ctx_chain_accumulate: the successor of a destroyed context simply stops waiting (no use of the freed event);
    a chained context refuses graph capture, and a capturing context cannot be chained."""
    from pyrad_amd import _native as nat, engine
    g = engine.layer_grid(1013.25, 640, 660, .001, False)
    sp = synthetic.SPECIES["co2"]
    mols = [dict(conc=4e-4, isotopologues=[dict(lines=synthetic.make_lines(92, 1500, g["eff_min"], g["eff_max"]),
                                                molmass=sp["molmass"], q_T=synthetic.q_value("co2", 280), q296=sp["q296"])])]
    a, b = nat.Context(0), nat.Context(0)
    try:
        La, Lb = (engine.ResidentLayer(c, 10.0, 280, 1013.25, 640, 660, mols, .001, False) for c in (a, b))
        Lb.enqueue(surface_T=288.0)
        ref = Lb.results()
        b.chain_accumulate(a)
        La.enqueue(surface_T=288.0); Lb.enqueue(surface_T=288.0)
        with pytest.raises(nat.LblError) as e:            # either end of a link: no capture
            b.capture(lambda: Lb.enqueue(surface_T=288.0))
        assert e.value.code == -6 and "chain" in str(e.value)
        with pytest.raises(nat.LblError) as e:
            a.capture(lambda: La.enqueue(surface_T=288.0))
        assert e.value.code == -6
        La.free()
        a.close()                                          # the predecessor goes first
        Lb.abs_coef.fill(0.0)
        Lb.enqueue(surface_T=288.0)                        # would wait on a destroyed event without the unlink
        assert np.array_equal(Lb.results()["abs_coef"], ref["abs_coef"])
        gb = Lb.capture_step(surface_T=288.0)              # and b is unchained now: capture works
        gb.launch()
        assert np.array_equal(Lb.results()["abs_coef"], ref["abs_coef"])
        gb.free(); Lb.free()
    finally:
        a.close(); b.close()


def test_sweeps_propagate_nan_in_both_arithmetics(ctx):
    """advisor, round 4: a NaN absorption coefficient (bad line data) must give a NaN transmittance and radiance, as
    np.exp(nan) does - the default arithmetic clamps its exponents with fmin / fmax, which drop a NaN - and a layer at the
    point n = 0 follows the reference's 0/0.  Both arithmetics of the sweeps, the column step and the fold."""
    n = 4096
    xs = np.full(n, 1e-22)
    bad = np.array([7, 100, 4095])
    xs[bad] = np.nan
    b = ctx.buffer(n).upload(xs)
    k, t, I, out = (ctx.buffer(n).fill(0.0) for _ in range(4))
    for ieee in (1, 0):
        ctx.set_option("sweep_ieee_divisions", ieee)
        try:
            ctx.layer_sweep_dev([b], [0], [4e-4], 1013.25, 296, 10.0, 0.0, 800.0, n, surface_T=288.0, abs_coef=k, trans=t, I_out=I)
            kk, tt, II = k.download(n), t.download(n), I.download(n)
            good = np.ones(n, bool); good[bad] = False
            assert np.all(np.isnan(kk[bad])) and np.all(np.isnan(tt[bad])) and np.all(np.isnan(II[bad])), ieee
            assert np.all(np.isfinite(kk[good])) and np.all(np.isfinite(tt[good])) and np.all(np.isfinite(II[good][1:])), ieee
            assert np.isnan(II[0])                      # nu = 0: B = 0 / (exp(0) - 1) = 0/0 in the reference too (pl:38-44)
            ctx.column_step_dev([dict(xsec=[b], iso_mol=[0], conc=[4e-4], P=1013.25, T=296, depth=10.0, abs_coef=None, trans=None)],
                                0.0, 800.0, n, out, surface_T=288.0)
            oo = out.download(n)
            assert np.all(np.isnan(oo[bad])) and np.all(np.isfinite(oo[good][1:])), ieee
        finally:
            ctx.set_option("sweep_ieee_divisions", 0)
    ctx.column_fold_dev([k], [296], [10.0], 0.0, 800.0, n, out, surface_T=288.0, trans=[t])
    oo, tt = out.download(n), t.download(n)
    assert np.all(np.isnan(oo[bad])) and np.all(np.isnan(tt[bad])) and np.all(np.isfinite(oo[good][1:]))
    for x in (b, k, t, I, out):
        x.free()


def test_fold_with_faint_incoming_radiance_and_nearly_transparent_layers(ctx):
    """Layer.transmission (cls:784-787), T I + (1 - T) B, at its extremes: incoming radiances from 1e-20 B to B behind layers
    of optical depth 1e-12 .. 1, through the fold's four-points-per-thread path, against the expression in extended
    precision.  What double precision allows here is set by T itself: exp(-tau) rounds with up to half an ulp of 1, which
    the factor (I - B) carries into the result - the bound below - and nothing beyond that may be lost (the kernel forms
    T I + ((1 - T) B) with one fma; B + T (I - B) would be cheaper and, as it happens, no worse than this bound)."""
    n = 4096
    rng = np.random.default_rng(5)
    nu = np.linspace(500.0, 800.0, n)
    h, c, kB = 6.62607004e-34, 299792458.0, 1.38064852e-23
    T = 250.0
    nul = nu.astype(np.longdouble)
    Bl = 2e8 * h * c * c * nul ** 3 / (np.exp(100 * h * c * nul / kB / T) - 1)
    tau = 10.0 ** rng.uniform(-12, 0, n)
    I_in = np.asarray(Bl, float) * 10.0 ** rng.uniform(-20, 0, n)
    depth = 1000.0
    k = tau / depth
    kb, ib, out = ctx.buffer(n).upload(k), ctx.buffer(n).upload(I_in), ctx.buffer(n)
    try:
        ctx.column_fold_dev([kb], [T], [depth], 500.0, 800.0, n, out, I_in=ib)
        got = out.download(n)
        tr = np.exp(-(k.astype(np.longdouble) * depth))
        want = tr * I_in.astype(np.longdouble) + (1 - tr) * Bl
        err = np.asarray(np.abs(got.astype(np.longdouble) - want) / want, float)
        bound = 1e-14 + 2.5e-16 * np.asarray(np.abs(I_in.astype(np.longdouble) - Bl) / want, float)
        assert np.all(np.isfinite(got)) and np.all(err <= bound), (float(np.max(err / bound)), int(np.argmax(err / bound)))
    finally:
        for x in (kb, ib, out):
            x.free()


def test_production_library_has_no_ablation_option(ctx):
    from pyrad_amd import _native as nat
    with pytest.raises(nat.LblError) as e:
        ctx.set_option("debug_ablate", 1)
    assert e.value.code == -1 and "unknown option" in str(e.value)
    # round 5: the superseded comparison kernels (variants 1, 2, 4) ship in diagnostic builds only
    for v in (1, 2, 4):
        with pytest.raises(nat.LblError) as e:
            ctx.set_option("accum_variant", v)
        assert e.value.code == -1 and "diagnostic builds only" in str(e.value)
    for v in (0, 3, 5):
        ctx.set_option("accum_variant", v)


def test_line_view_bounds_cannot_overflow(ctx):
    from pyrad_amd import _native as nat
    L = ctx.lines(synthetic.make_lines(93, 100, 600.0, 700.0))
    h = nat._P()
    assert ctx.lib.lbl_lines_view(L.h, 50, 2**63 - 1, nat.C.byref(h)) == -1
    assert ctx.lib.lbl_lines_view(L.h, 2**62, 2**62, nat.C.byref(h)) == -1
    L.free()


def test_asynchronous_download_lands_behind_the_work_before_it(ctx):
    """lbl_buffer_download_async / lbl_download_wait: pieces of a buffer leave for page-locked memory behind the kernels
    enqueued before each call and beside those enqueued after it; lbl_sync waits for them too; bad ranges are refused."""
    from pyrad_amd import _native as nat
    n = 300001
    rng = np.random.default_rng(11)
    a = rng.random(n)
    buf = ctx.buffer(n).upload(a)
    total = ctx.buffer(n)
    host = ctx.host_array(n)
    host[:] = -1.0
    for lo, cnt in ((0, 100000), (100000, 100000), (200000, 100001)):
        ctx.sum_dev([buf, buf], n, total)                      # (a kernel on the context stream before every piece)
        total.download_async(host, cnt, lo, lo)
        buf.fill(0.0) if lo == 200000 else None                # later work on ANOTHER buffer does not disturb the copies
    ctx.download_wait()
    assert np.array_equal(host, a + a)
    host[:] = -1.0
    total.download_async(host, n)
    ctx.sync()                                                 # (lbl_sync covers the copy stream)
    assert np.array_equal(host, a + a)
    with pytest.raises(nat.LblError):
        total.download_async(host, n, 1)                       # range past the end of the buffer
    with pytest.raises(ValueError):
        total.download_async(host[:10], 11)
    total.free(); buf.free()
