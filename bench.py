#!/usr/bin/env python3
"""bench.py — line x grid-point evaluations per second of the MI355X line-by-line engine.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over one batch of synthetic input already resident in
HBM: per-line preparation (K1), owner-computes accumulation of every line onto the
wavenumber grid with the absorption-coefficient / transmittance / Planck-emission sweep folded
into its output stage (K2), and, for N > 1, the single RCCL all-gather of the per-rank spectrum
shards.

Workload (BASELINE.json): N = 1 -> configs[2], the CO2+H2O+CH4 mixed cell, 100-2500 cm^-1 at
0.001 cm^-1 (2.4e6 grid points, W = 5000), 3 x 131,072 seeded synthetic lines, 1013.25 mbar,
296 K (SURVEY.md §8d "C3") - the configuration the metric's 1/2/4/8 ladder is quoted on.
N > 1 -> configs[3]: the SAME fixed workload with the wavenumber grid sharded by contiguous
range across the N ranks ("scaling": "strong"); shard boundaries are cost-balanced from the line
positions.  `--workload C1|C2|C5` select the other configurations, `--weak` grows the grid with N
instead (every rank owns a whole C-sized grid).

`python bench.py --gpus N` with N > 1 and no launcher environment starts its own N ranks (fresh
child processes, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU
and relays rank 0's line; under torch.distributed.run it uses the ranks it is given.

Rank 0 prints ONE JSON line.  `value` = exact (line, grid-point) contributions of the
reference's scatter loop (pyradClasses.py:392-400) summed over all ranks and steps / the
max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
FP64_VALU_PEAK_INSTR = 256 * 4 * 16 * 2.4e9     # fp64 lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz
FP64_INSTR_PER_EVAL = 5.0        # running-fraction Lorentz inner loop: add, fma, mul, fma, mul


def build_workload(workload: str, world: int):
    from pyrad_amd import synthetic
    if workload == "C2":
        rmin, rmax = 500, 500 + 400 * world
        cfg = synthetic.config_c2(n_lines=65536 * world, range_min=rmin, range_max=rmax, seed=2)
        desc = "CO2 %d-%d cm^-1 @0.001, %d lines, 1013.25 mbar, 296 K, 10 cm (C2 shape%s)" % (
            rmin, rmax, 65536 * world, "" if world == 1 else ", grid sharded x%d" % world)
    elif workload == "C3":
        rmin, rmax = 100, 100 + 2400 * world
        cfg = synthetic.config_c3(n_lines=131072 * world, range_min=rmin, range_max=rmax)
        desc = "CO2+H2O+CH4 %d-%d cm^-1 @0.001, 3x%d lines, 1013.25 mbar, 296 K (C3 shape)" % (rmin, rmax, 131072 * world)
    elif workload == "C5":
        cfg = synthetic.config_c5(n_layers=30, n_lines=131072, range_min=100, range_max=100 + 2400 * world)
        desc = ("30-layer column H2O+CO2+O3 100-%d cm^-1 @0.001, 3x131072 lines per layer window, "
                "P 1013->10 mbar (C5 shape)" % (100 + 2400 * world))
    elif workload == "C1":
        cfg = synthetic.config_c1()
        desc = "CO2 600-700 cm^-1 @0.01, 4096 lines (C1, the reference's own CPU-runnable case)"
    else:
        raise SystemExit("unknown workload %s" % workload)
    return cfg, desc


def molecules_of(cfg):
    """layer config dict -> ResidentLayer molecule descriptions (host logic only)."""
    from pyrad_amd import synthetic
    from pyrad_amd.model import concentration_from_kwargs
    mols = []
    for mol in cfg["molecules"]:
        sp = synthetic.SPECIES[mol["species"]]
        mols.append(dict(conc=concentration_from_kwargs(**mol["conc"]),
                         isotopologues=[dict(lines=mol["lines"], molmass=sp["molmass"],
                                             q_T=synthetic.q_value(mol["species"], cfg["T"]), q296=sp["q296"])]))
    return mols


def cpu_baseline(layer_cfgs, workload, seconds_target=15.0):
    """The reference's hot loop on ONE host core (the reference is single-threaded), SURVEY.md §8(d): three
    restatements - the faithful Python/NumPy scalar loop (kind "port": the reported value), the vectorised NumPy
    form and the plain-C port - ALL timed on the SAME seeded sample of the workload's lines, drawn uniformly from
    every line list of every layer and sized so that the Python leg takes about ``seconds_target``; and the C and
    vectorised legs once more at §8(d)'s extent: the whole workload for C1 and C2, a seeded 1/16 of every line
    list for C3 and C5 (linear in the eval count, stated as such)."""
    from oracle import pyrad_oracle as orc
    from oracle import c_oracle
    from pyrad_amd import synthetic
    jobs = []
    for cfg in layer_cfgs:
        grid = orc.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"],
                              cfg.get("dynamic_resolution", True))
        for mol in cfg["raw_molecules"]:
            sp = synthetic.SPECIES[mol["species"]]
            lines = orc.select_window(mol["lines"], grid["eff_min"], grid["eff_max"])
            jobs.append(dict(lines=lines, args=(cfg["T"], cfg["P"], orc.concentration(**mol["conc"]), sp["molmass"],
                                                synthetic.q_value(mol["species"], cfg["T"]), sp["q296"], grid)))

    def subset(fraction, seed):
        out = []
        for i, j in enumerate(jobs):
            n = len(j["lines"]["nu"])
            if fraction >= 1.0:
                out.append(j["lines"])
                continue
            rng = np.random.default_rng(seed + i)
            pick = np.sort(rng.choice(n, size=max(1, int(round(n * fraction))), replace=False)) if n else np.zeros(0, np.int64)
            out.append({k: v[pick] for k, v in j["lines"].items()})
        return out

    def evals_of(sub):
        total = 0
        for lines, j in zip(sub, jobs):
            T, P, conc, molmass, qT, q296, grid = j["args"]
            lq = orc.line_quantities(lines, T, P, conc, molmass, grid["range_min"], grid["resolution"])
            total += int(orc.eval_count(lq["index"], grid["W"], grid["n_work"]))
        return total

    def run(kind, sub):
        t0 = time.perf_counter()
        for lines, j in zip(sub, jobs):
            if kind == "python":
                orc.create_cross_section_scalar(lines, *j["args"], regrid=False)
            elif kind == "numpy":
                orc.create_cross_section(lines, *j["args"], regrid=False)
            else:
                c_oracle.create_cross_section_work(lines, *j["args"])
        return time.perf_counter() - t0

    c_oracle.load()
    e_total = evals_of([j["lines"] for j in jobs])
    n_total = sum(len(j["lines"]["nu"]) for j in jobs)
    # calibrate the Python leg on a sliver, then size the common sample
    f0 = min(1.0, 2.0e6 / max(e_total, 1))
    cal = subset(f0, 100)
    rate0 = evals_of(cal) / max(run("python", cal), 1e-9)
    f = min(1.0, seconds_target * rate0 / max(e_total, 1))
    S = subset(f, 0)
    e_S, n_S = evals_of(S), sum(len(x["nu"]) for x in S)
    t_py, t_np, t_c = run("python", S), run("numpy", S), run("c", S)
    f_big = 1.0 if workload in ("C1", "C2") else 1.0 / 16.0
    B = subset(f_big, 0) if f_big != f else S
    e_B, n_B = evals_of(B), sum(len(x["nu"]) for x in B)
    tb_np, tb_c = run("numpy", B), run("c", B)
    what_B = ("the WHOLE workload" if f_big >= 1.0 else
              "a seeded 1/16 of every line list (SURVEY.md 8d; the whole workload extrapolates linearly in the eval count: %.3g evals)" % e_total)
    return {
        "value": e_S / t_py, "unit": "line*gridpoint evals/s", "cores": 1, "kind": "port",
        "sample": "%d of the workload's %d lines (fraction %.4g of every line list of every layer, seed 0), %d of %d evals; "
                  "faithful Python/NumPy scalar restatement of pyradClasses.py:361-400 (oracle.create_cross_section_scalar) in %.1f s; "
                  "host has %d logical cores, 1 used (the reference is single-threaded)" % (
                      n_S, n_total, f, e_S, e_total, t_py, os.cpu_count() or 0),
        "vectorised_value": e_S / t_np,
        "vectorised_sample": "vectorised NumPy restatement (oracle.create_cross_section, 1 core) on the SAME sample: %.2f s" % t_np,
        "c_port_value": e_S / t_c,
        "c_port_sample": "plain-C restatement (oracle/lbl_oracle.c, gcc -O2, 1 core) on the SAME sample: %.2f s" % t_c,
        "at_survey_extent": {
            "what": what_B, "lines": n_B, "evals": e_B,
            "c_port_value": e_B / tb_c, "c_port_seconds": tb_c,
            "vectorised_value": e_B / tb_np, "vectorised_seconds": tb_np,
            "whole_workload_seconds_c_port": e_total / (e_B / tb_c),
            "whole_workload_seconds_python": e_total / (e_S / t_py),
        },
    }


def api_path(cfg, reps=5):
    """The drop-in route: pyrad_amd.model's Layer -> addMolecule -> getAbsCoef -> transmission on the
    bench workload, host arrays in and out (PCIe included).  ms_per_call = one getAbsCoef after
    changeTemperature has invalidated every cross section (line prep + accumulate + fused sweep on
    the resident line lists, then ONE download, the absorption coefficient)."""
    from pyrad_amd import model, data, settings, engine
    keep = (settings.RES_MULTIPLIER, data._source, model.Layer.hasAtmosphere)
    try:
        settings.set_resolution_multiplier(cfg["base_resolution"] / .01)
        data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in cfg["molecules"]}))
        model.Layer.hasAtmosphere = False
        t0 = time.perf_counter()
        engine.get_engine()                                 # context, stream, page-locked staging pool: once per process
        t_engine = time.perf_counter() - t0
        t0 = time.perf_counter()
        layer = model.Layer(cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"],
                            dynamicResolution=cfg.get("dynamic_resolution", True))
        for m in cfg["molecules"]:
            layer.addMolecule(m["species"], **m["conc"])
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        k = model.getAbsCoef(layer)                         # first call: uploads the lines, builds the schedule
        t_first = time.perf_counter() - t0
        evals = sum(engine.eval_count(iso._lines["nu"], cfg["range_min"], layer.resolution, layer._grid()["W"],
                                      layer._grid()["n_work"]) for m in layer for iso in m)
        t_call, t_trans = [], []
        surf = layer.planck(288)
        for _ in range(reps):
            layer.changeTemperature(cfg["T"])               # marks every cross section dirty (cls:741-743)
            t0 = time.perf_counter()
            k = model.getAbsCoef(layer)
            t_call.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            spec = layer.transmission(surf)
            t_trans.append(time.perf_counter() - t0)
        ms_call = 1e3 * float(np.median(t_call))
        # re-windowing (pyradClasses.py:734-752 -> resetData, cls:45-56): the mutator re-reads the lines of the new window,
        # the next getter computes on it - new line selection (a view of the resident list), a new dispatch schedule
        # (built on the device, in stream), line prep, accumulate, sweep, one download
        t_mut_p, t_get_p, t_mut_r, t_get_r = [], [], [], []
        P0, r0 = cfg["P"], (cfg["range_min"], cfg["range_max"])
        width = r0[1] - r0[0]
        for i in range(reps):
            P_new = P0 * (0.9 - 0.02 * i)
            t0 = time.perf_counter(); layer.changePressure(P_new); t_mut_p.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); k2 = model.getAbsCoef(layer); t_get_p.append(time.perf_counter() - t0)
        layer.changePressure(P0)
        for i in range(reps):
            a, b = r0[0] + width * 0.02 * (i + 1), r0[1] - width * 0.02 * (i + 1)
            t0 = time.perf_counter(); layer.changeRange(a, b); t_mut_r.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); k2 = model.getAbsCoef(layer); t_get_r.append(time.perf_counter() - t0)
        layer.changeRange(*r0)
        med = lambda v: 1e3 * float(np.median(v))
        return {"ms_per_call": ms_call, "evals_per_s": evals / (ms_call * 1e-3),
                "ms_transmission": 1e3 * float(np.median(t_trans)),
                "ms_change_pressure": med(t_get_p), "ms_change_pressure_mutator": med(t_mut_p),
                "ms_change_range": med(t_get_r), "ms_change_range_mutator": med(t_mut_r),
                "rewindow_what": "getAbsCoef AFTER layer.changePressure / layer.changeRange (each to a window not seen before: new "
                                 "line selection, new dispatch schedule, recompute, one download; finite %s); *_mutator = the "
                                 "changePressure / changeRange call itself (host: re-reading the window's lines)" % bool(np.isfinite(k2).all()),
                "ms_engine_create": 1e3 * t_engine,
                "ms_build_layer": 1e3 * t_build, "ms_first_call": 1e3 * t_first, "bytes_downloaded_per_call": 8 * int(k.size),
                "what": "model.getAbsCoef(layer) after layer.changeTemperature (recompute on resident line lists + download "
                        "of the absorption coefficient); ms_transmission = layer.transmission(host spectrum): upload, fold "
                        "with the resident transmittance, download; medians of %d; checks: %d points, finite %s" % (
                            reps, spec.size, bool(np.isfinite(spec).all() and np.isfinite(k).all()))}
    finally:
        settings.set_resolution_multiplier(keep[0])
        data.set_source(keep[1])
        model.Layer.hasAtmosphere = keep[2]
        engine.shutdown()


def api_path_column(cfg, reps=5):
    """The drop-in route for the column: pyrad_amd.model's Atmosphere -> addLayer -> addMolecule ->
    Atmosphere.transmission(surfaceTemperature) on the bench column, host array out.  ms_per_call = one
    transmission after every layer's changeTemperature has invalidated its cross sections: all layers' line
    lists go through ONE batched accumulate launch sequence and one column-step kernel, then one download."""
    from pyrad_amd import model, data, settings, engine
    keep = (settings.RES_MULTIPLIER, data._source, model.Layer.hasAtmosphere)
    try:
        c0 = cfg["layers"][0]
        settings.set_resolution_multiplier(c0["base_resolution"] / .01)
        data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in c0["molecules"]}))
        model.Layer.hasAtmosphere = False
        t0 = time.perf_counter()
        engine.get_engine()
        t_engine = time.perf_counter() - t0
        t0 = time.perf_counter()
        atm = model.Atmosphere("bench column")
        for c in cfg["layers"]:
            layer = atm.addLayer(c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], name=c["name"],
                                 dynamicResolution=c.get("dynamic_resolution", True))
            for m in c["molecules"]:
                layer.addMolecule(m["species"], **m["conc"])
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        spec = atm.transmission(surfaceTemperature=cfg["surface_T"])       # first call: uploads the lines, builds the schedules
        t_first = time.perf_counter() - t0
        evals = sum(engine.eval_count(iso._lines["nu"], L.rangeMin, L.resolution, L._grid()["W"], L._grid()["n_work"])
                    for L in atm for m in L for iso in m)
        def timed_call():
            for L in atm:
                L.changeTemperature(L.T)                    # marks every cross section of the layer dirty (cls:741-743)
            t0 = time.perf_counter()
            out = atm.transmission(surfaceTemperature=cfg["surface_T"])
            return time.perf_counter() - t0, out
        # The device has idled while the host counted the evaluations above: the first calls run at lower clocks (6.1, 5.6,
        # 5.6, 5.5, 5.4 ms measured back to back).  They are kept as ms_calls_cold; ms_per_call is the median of `reps`
        # calls after a quarter of a second of the same calls, the preconditioning the timed loop of the step gets too.
        t_cold = [timed_call()[0] for _ in range(3)]
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < 0.25:
            timed_call()
        t_call = []
        for _ in range(reps):
            t, spec = timed_call()
            t_call.append(t)
        ms_call = 1e3 * float(np.median(t_call))
        # re-windowing every layer of the column (pyradClasses.py:745-752): 1 % lower pressures, then back
        t_mut, t_get = [], []
        for f in (0.99, 1.0):
            t0 = time.perf_counter()
            for L, c in zip(atm, cfg["layers"]):
                L.changePressure(c["P"] * f)
            t_mut.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            spec = atm.transmission(surfaceTemperature=cfg["surface_T"])
            t_get.append(time.perf_counter() - t0)
        return {"ms_per_call": ms_call, "ms_calls": [round(1e3 * t, 4) for t in t_call],
                "ms_calls_cold": [round(1e3 * t, 4) for t in t_cold],
                "evals_per_s": evals / (ms_call * 1e-3), "ms_build_atmosphere": 1e3 * t_build,
                "ms_engine_create": 1e3 * t_engine,
                "ms_change_pressure": 1e3 * float(np.median(t_get)), "ms_change_pressure_mutator": 1e3 * float(np.median(t_mut)),
                "rewindow_what": "Atmosphere.transmission AFTER changePressure on every layer (new windows: new line selections, "
                                 "new dispatch schedules built on the device, recompute, one download); *_mutator = the %d "
                                 "changePressure calls themselves" % len(cfg["layers"]),
                "ms_first_call": 1e3 * t_first, "bytes_downloaded_per_call": 8 * int(spec.size),
                "what": "model.Atmosphere.transmission(surfaceTemperature) after changeTemperature on all %d layers "
                        "(recompute on resident line lists, one accumulate job per layer, the fold in four pieces with the outgoing "
                        "spectrum downloaded beside the next piece); median of %d consecutive calls (ms_calls) after 0.25 s of the same "
                        "calls; ms_calls_cold = the first three calls after the idle set-up, at the clocks the device had dropped to; "
                        "checks: %d points, finite %s" % (len(cfg["layers"]), reps, spec.size, bool(np.isfinite(spec).all()))}
    finally:
        settings.set_resolution_multiplier(keep[0])
        data.set_source(keep[1])
        model.Layer.hasAtmosphere = keep[2]
        engine.shutdown()


def choose_step(step_arg, n_points, n_lists, n_layers, shard_world, unfused=False, variant=None):
    """--step auto: the merged step unless the cell's per-list accumulate launch is at most about two rounds of workgroups
    (measured on shards of the 100-2500 cm^-1 cell, per-list / merged: of 8 0.0575 / 0.0589 ms, of 4 0.0943-0.0961 /
    0.0970-0.0975, of 2 0.1658-0.1680 / 0.1559-0.1577, the whole cell 0.306 / 0.260)."""
    few_rounds = float(n_points) * n_lists / shard_world <= 2.1 * 4.0 * 256 * 1024
    return (step_arg == "merged" or (step_arg == "auto" and not (few_rounds and n_lists > n_layers))) \
        and not unfused and variant in (None, 3, 5)


def workload_leg(name, device, steps, warmup, step_arg="auto", accuracy="exact", want_api=True, shards="auto",
                 ctx=None, comm=None, world=1, rank=0, red=None, cfg_desc=None):
    """One BASELINE configuration beside the line's own (round-5 verdict, item 2: C1, C2 and C5 had only ever been timed by
    the builder): the resident step of that workload, `steps` steps between stream syncs after `warmup` steps and a tenth
    of a second of preconditioning, then a few steps with every kernel class bracketed by events.  world > 1 (C5 under
    `--gpus N`): every rank computes its contiguous grid range of all layers (the fold is independent per grid point) and
    ONE all-gather per step collects the outgoing spectrum (in stream: a step's spectrum is complete before the next step
    starts); ms_per_step is then the max over ranks between RCCL barriers, and the breakdown times the step's kernels and
    its all-gather separately.  Untimed-region style: not the line's value."""
    from pyrad_amd import _native as nat, engine
    cfg, desc = cfg_desc if cfg_desc is not None else build_workload(name, 1)
    is_column = name == "C5"
    if is_column:
        layer_cfgs = [dict(c, molecules=molecules_of(c)) for c in cfg["layers"]]
    else:
        layer_cfgs = [dict(cfg, molecules=molecules_of(cfg))]
    own_ctx = ctx is None
    if own_ctx:
        ctx = nat.Context(device)
        if accuracy == "budget":
            ctx.set_option("accuracy", 1)
    shard, shard_choice = None, "none"
    if world > 1:
        shard, shard_choice = engine.choose_shards(layer_cfgs, world, rank, shards)
    t0 = time.perf_counter()
    if is_column:
        L = engine.ResidentColumn(ctx, layer_cfgs, cfg["surface_T"], shard=shard)
        g = L.layers[0].g
        n_lists, n_layers = len(L.jobs), len(L.layers)
    else:
        c = layer_cfgs[0]
        L = engine.ResidentLayer(ctx, c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], c["molecules"],
                                 c["base_resolution"], c.get("dynamic_resolution", True), shard=shard)
        g = L.g
        n_lists, n_layers = len(L.jobs), 1
    merged = choose_step(step_arg, g["n_work"], n_lists, n_layers, world)
    kw = dict(layer_arrays=False, merged=merged) if is_column else dict(surface_T=288.0, merged=merged)

    def one():
        L.enqueue(**kw)
        if comm is not None:
            if is_column:
                L.enqueue_allgather(comm)
            else:
                L.enqueue_allgather(comm, (L.abs_coef,))

    def barrier():
        ctx.sync()
        if comm is not None:
            comm.fence_dev(-1)
            ctx.sync()
            comm.allgather_dev(red, rank, 1, red)
            ctx.sync()

    def max_over_ranks(v):
        if comm is None:
            return float(v), [float(v)]
        red.upload(np.array([v], dtype=np.float64), offset=rank)
        comm.allgather_dev(red, rank, 1, red)
        a = red.download(world)
        return float(a.max()), [float(x) for x in a]

    one()
    barrier()
    t_setup = time.perf_counter() - t0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.1:
        for _ in range(5):
            one()
        ctx.sync()
    for _ in range(warmup):
        one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    barrier()
    t_step, t_by_rank = max_over_ranks((time.perf_counter() - t0) / steps)
    classes = ["line_prep", "xsec_accumulate", "layer_sweep", "column_sweep"] + (["allgather"] if comm is not None else [])
    n_extra = 3
    ctx.profile_enable(classes)
    ctx.profile_reset()
    for _ in range(n_extra):
        one()
    barrier()
    prof = ctx.profile_read()
    ctx.profile_enable(False)
    kernel_ms = {k: prof[k][1] / n_extra for k in classes if prof[k][0]}
    evals_local = float(L.evals)
    evals_total = evals_local
    if comm is not None:
        red.upload(np.array([evals_local], dtype=np.float64), offset=rank)
        comm.allgather_dev(red, rank, 1, red)
        evals_total = float(red.download(world).sum())
    pts = L.count if L.plan is not None else g["n_work"]
    out = {"workload": desc, "step": "merged" if merged else "per-list", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": t_step * 1e3, "evals_per_step": evals_total, "evals_per_s": evals_total / t_step,
           "kernel_ms_per_step": kernel_ms, "grid_points_per_gpu": int(pts), "lines_per_gpu": int(L.n_lines),
           "xsec_accumulate_launches_per_step": prof["xsec_accumulate"][0] / n_extra, "setup_s": t_setup, "accuracy": accuracy,
           "what": "%d steps of the resident step between stream syncs (max over ranks) after %d warm-up steps and 0.1 s of the same "
                   "steps; kernel_ms_per_step from %d further steps with every kernel class bracketed by HIP events (a pair of event "
                   "records costs the stream a few microseconds: the parts can sum to more than ms_per_step); not the line's value"
                   % (steps, warmup, n_extra)}
    if is_column and merged and kernel_ms.get("column_sweep") and kernel_ms.get("line_prep"):
        t_fold, t_k1 = kernel_ms["column_sweep"] * 1e-3, kernel_ms["line_prep"] * 1e-3
        b_fold = 8.0 * pts * (n_layers + 1)                        # one absorption coefficient per layer read, the outgoing spectrum written
        b_k1 = 128.0 * L.n_lines                                   # per line: 56 B of HITRAN fields + 4 B of the merged-order map read, 68 B of records written
        out["fold"] = {"kernel": "column_step_kernel", "algorithmic_bytes_per_launch": b_fold, "avg_launch_ms": t_fold * 1e3,
                       "hbm_frac": b_fold / t_fold / 1e9 / HBM_PEAK_GBS}
        out["line_prep"] = {"kernel": "line_prep_merged_kernel", "algorithmic_bytes_per_launch": b_k1, "avg_launch_ms": t_k1 * 1e3,
                            "hbm_frac": b_k1 / t_k1 / 1e9 / HBM_PEAK_GBS}
    if comm is not None:
        # where the sharded step's time goes: its kernels without the collective, and the collective alone, in stream
        n_b = max(5, min(20, steps))
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_b):
            L.enqueue(**kw)
        barrier()
        t_k, k_by_rank = max_over_ranks((time.perf_counter() - t0) / n_b)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_b):
            if is_column:
                L.enqueue_allgather(comm)
            else:
                L.enqueue_allgather(comm, (L.abs_coef,))
        barrier()
        t_g, _ = max_over_ranks((time.perf_counter() - t0) / n_b)
        recv = (world - 1) * L.S * 8.0
        out["config"] = {"shards": shard_choice, "shard_bounds": None if L.plan is None else [list(b) for b in L.plan.bounds],
                         "allgather": "in stream, one per step: the outgoing spectrum" if is_column else "in stream, one per step: the absorption coefficient"}
        out["sharded_step_breakdown"] = {"kernels_only_ms_per_step": t_k * 1e3, "kernels_only_ms_by_rank": [round(v * 1e3, 4) for v in k_by_rank],
                                         "step_ms_by_rank": [round(v * 1e3, 4) for v in t_by_rank],
                                         "allgather_alone_ms_per_step": t_g * 1e3, "allgather_bytes_received_per_rank_per_step": recv,
                                         "allgather_alone_GBps_per_rank": (recv / t_g / 1e9) if t_g > 0 else None,
                                         "what": "two short passes after the leg's timed steps, wall clock between barriers, max over ranks: the "
                                                 "step's kernels with no all-gather, and its all-gather alone in stream"}
    L.free()
    if own_ctx:
        ctx.close()
    if want_api and world == 1 and rank == 0:
        out["api_path"] = api_path_column(cfg) if is_column else api_path(cfg)
    return out


class _StdoutToStderr:
    """RCCL prints a version banner on stdout when a communicator is created; rank 0 must print
    exactly one JSON line there, so file descriptor 1 points at stderr while RCCL initialises."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class _Contexts:
    """The contexts of this rank: one per step in flight (each has its own HIP stream, scratch arenas and
    resident line lists); options and profiling apply to all of them, profile reads are summed."""

    def __init__(self, ctxs):
        self.all = list(ctxs)
        self.first = self.all[0]

    def set_option(self, key, value):
        for c in self.all:
            c.set_option(key, value)

    def sync(self):
        for c in self.all:
            c.sync()

    def profile_enable(self, what):
        for c in self.all:
            c.profile_enable(what)

    def profile_reset(self):
        for c in self.all:
            c.profile_reset()

    def profile_reserve(self, n):
        for c in self.all:
            c.profile_reserve(n)

    def profile_read(self):
        out = {}
        for c in self.all:
            for name, (n_, ms_) in c.profile_read().items():
                a = out.get(name, (0, 0.0))
                out[name] = (a[0] + n_, a[1] + ms_)
        return out

    def close(self):
        for c in self.all:
            c.close()


def partial_round(point_lists: float, n_cu: int = 256) -> bool:
    """Is a step's accumulate launch at most about one round of workgroups (1024 points of one line list each,
    4 resident per CU)?  879 for 1024 slots at 8 shards of C3, 782 for C2."""
    return point_lists <= 1.05 * 4.0 * n_cu * 1024


def steps_in_flight(requested: str, sharded: bool) -> int:
    """How many independent steps a rank keeps in flight (each on a HIP stream of its own).  A shard's
    accumulate launch is a partial round of workgroups or a few rounds: its tail runs with 1-3 wavefronts per
    SIMD at 50-95 % of the fp64 rate, and K1 / the sweep leave the VALU idle; a second step fills both.
    Measured on one GPU, kernels only: a shard of 8 of C3 0.063 -> 0.052 -> 0.049 ms per step with 1 / 2 / 3
    in flight, of 4 0.104 -> 0.090 -> 0.086, of 2 0.176 -> 0.159 -> 0.158, of 8 of the column 0.82 -> 0.72;
    with the all-gather pipeline beside them 2 beats 3 (0.0624 / 0.0638 at 8, 0.097 / 0.101 at 4).
    auto: 2 for a shard; 1 for an unsharded grid, so that an N = 1 line times every kernel alone on the chip and
    agrees with its rocprofv3 summary (the profiler does not let launches of different streams overlap the
    way they do unobserved); what more steps in flight give a small unsharded cell (C2 0.066 -> 0.043 ms,
    C1 0.0185 -> 0.0087 with three) is reported by an extra leg of the line, `in_flight_leg`.
    Round 3: auto is 1 everywhere.  Two contexts sharing one communicator have only ever run with one rank
    (no multi-GPU box is available to the build), so the default N > 1 line takes the simple path - one step in
    flight, one collective per step - and 2 / 3 in flight stay opt-in (--in-flight 2)."""
    if requested != "auto":
        return max(1, min(3, int(requested)))
    return 1


def gather_batch(requested: str, batchable: bool, steps_per_set: int = 0) -> int:
    """Steps whose shards leave in ONE all-gather.  At 8 ranks a C3 shard is 2.4 MB and a step 0.05 ms: a
    collective per step is dominated by its start-up (a ring of 7 hops) and runs far below the links' rate;
    the shards of B consecutive steps of a resident set are staged side by side in a batch buffer (rank r,
    step b at [(r B + b) S, +S)) and sent as one collective of B S doubles per rank - fewer, larger
    collectives, same bytes.  Two batch buffers per set: the gather of one batch overlaps the steps that fill
    the other.  With ONE rank forced through the communicator a shard of 8 of C3 steps in 0.0626 ms with a
    collective per step and 0.0537 / 0.0528 / 0.0522 with B = 2 / 4 / 8 (kernels alone: 0.0523).
    Round 3: auto is 1 (a collective per step: the path that has run, and the one whose per-step result is usable
    as soon as its own gather is done); "fit" picks the B in 3..8 that leaves the smallest partly filled batch at
    the end of the timed region (a flush sends the whole buffer), larger B on a tie - 5 for 20 steps over two
    resident sets; an integer forces B."""
    if not batchable or requested == "auto":
        return 1
    if requested != "fit":
        return max(1, min(16, int(requested)))
    if steps_per_set < 3:
        return max(1, steps_per_set)
    best = None
    for b in range(3, 9):
        waste = -(-steps_per_set // b) * b - steps_per_set
        if best is None or waste <= best[0]:
            best = (waste, b)
    return best[1]


class _Batch:
    """The two batch buffers of one resident set: world * B * S doubles each, gathered in place."""

    def __init__(self, ctx, comm, rank, world, S, B, slots):
        self.comm, self.rank, self.S, self.B, self.slots = comm, rank, S, B, slots
        self.bufs = [ctx.buffer(world * B * S).fill(0.0) for _ in range(2)]
        self.cur, self.fill, self.sent = 0, 0, 0

    def before_step(self):
        if self.fill == 0:
            self.comm.fence_dev(self.slots[self.cur])     # the gather that last read this buffer (two batches ago) is done

    def stage(self, src, src_offset):
        self.bufs[self.cur].stage_from_dev(src, src_offset, self.S, dst_offset=(self.rank * self.B + self.fill) * self.S)
        self.fill += 1
        if self.fill == self.B:
            self.send()

    def send(self, overlap=True):
        """one all-gather for the whole batch buffer (a partial batch at a flush sends the full buffer too)"""
        buf = self.bufs[self.cur]
        self.comm.allgather_dev(buf, self.rank * self.B * self.S, self.B * self.S, buf,
                                overlap_slot=self.slots[self.cur] if overlap else None)
        self.cur ^= 1
        self.fill = 0
        self.sent += 1

    def flush(self):
        if self.fill:
            self.send()

    def free(self):
        for b in self.bufs:
            b.free()


def in_flight_leg(cfg, n_flight=3, steps=200, chained=False, merged=True):
    """Throughput of the same resident cell with n_flight independent steps in flight, each on a context (HIP
    stream) of its own, dealt round-robin: an extra leg for cells whose accumulate launch is a partial round
    of workgroups.  Same kernels, same results; not the line's `value`."""
    from pyrad_amd import _native as nat, engine
    mols = molecules_of(cfg)
    ctxs = [nat.Context(int(os.environ.get("LOCAL_RANK", "0"))) for _ in range(n_flight)]
    layers = [engine.ResidentLayer(c, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                   cfg["base_resolution"], cfg.get("dynamic_resolution", True)) for c in ctxs]
    if chained:           # software pipeline: the accumulate kernels run one after another, line prep and sweeps beside them
        for i, c in enumerate(ctxs):
            c.chain_accumulate(ctxs[i - 1])

    def run(n):
        for k in range(n):
            layers[k % n_flight].enqueue(surface_T=288.0, merged=merged)
        for c in ctxs:
            c.sync()
    run(n_flight)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        run(30)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(steps)
        best = min(best, (time.perf_counter() - t0) / steps)
    evals = float(layers[0].evals)
    for L in layers:
        L.free()
    for c in ctxs:
        c.close()
    return {"steps_in_flight": n_flight, "ms_per_step": best * 1e3, "evals_per_s": evals / best, "chained": bool(chained),
            "what": "the same cell with %d independent steps in flight on %d HIP streams (contexts), %d steps, best of 3: "
                    "further steps fill the SIMDs that a partial round of workgroups leaves underused%s" % (
                        n_flight, n_flight, steps,
                        "" if not chained else "; chained (lbl_ctx_chain_accumulate): every step's accumulate kernels wait for the "
                        "previous step's, so they run alone on the chip, one after another, with the line prep of the next step "
                        "and the sweep of the previous one beside them")}


# ------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher starts N ranks itself
# ------------------------------------------------------------------------------------------
def _free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_plan(n_ranks: int, argv, environ, port: int | None = None):
    """[(argv, env)] for the N child ranks: the torchrun contract (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT), one rendezvous key for the file exchange of the RCCL unique id, and
    the dmabuf IPC setting RCCL needs on this driver.  Pure host logic (tests/test_bench_launcher_cpu.py)."""
    port = port or _free_port()
    base = dict(environ)
    base.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                PYRAD_RENDEZVOUS_KEY="bench_%d_%d" % (os.getpid(), port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    plan = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        plan.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return plan


def run_ranks(plan, timeout_s: float = 3000.0) -> int:
    """Start the ranks, wait for all of them, relay rank 0's stdout (the one JSON line).  A rank that
    fails ends the others (they would wait for it in the rendezvous) and its status is returned."""
    import subprocess
    procs = []
    for r, (argv, env) in enumerate(plan):
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    t0 = time.time()
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
            if bad or time.time() - t0 > timeout_s:
                rc = bad[0] if bad else 124
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                break
            time.sleep(0.05)
        out = procs[0].stdout.read().decode() if procs[0].stdout else ""
        for p in procs:
            try:
                p.wait(30)
            except Exception:
                p.kill()
        rc = rc or next((p.returncode for p in procs if p.returncode), 0)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C5"],
                    help="BASELINE configuration (default C3: the mixed cell the metric is quoted on)")
    ap.add_argument("--variant", type=int, default=None, help="accumulate kernel variant 0..5 (default: library default 5, far-field series)")
    ap.add_argument("--points-per-lane", type=int, default=None)
    ap.add_argument("--line-split", type=int, default=None)
    ap.add_argument("--tile-order", type=int, default=None)
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    ap.add_argument("--precondition-seconds", type=float, default=0.3,
                    help="setup runs the step for this long before the warm-up steps so that the GPU is at its sustained clocks")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="HIP events around the dominant kernel in every n-th timed step only (default 4; 1 = every step)")
    ap.add_argument("--column-layer-arrays", type=int, default=0,
                    help="C5: 1 also writes every layer's absorption coefficient and transmittance arrays (default 0: "
                         "the column step keeps them in registers and writes the outgoing spectrum only)")
    ap.add_argument("--longest-first", type=int, default=None, help="0: positional tile order, 1: longest-first worklist, 2: snake order, 3 (default): bin-packed per CU on single-round launches")
    ap.add_argument("--scale", type=int, default=1, help="experiment: widen the per-GPU range and line count by this factor")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: grow the grid with N (every rank owns one whole C-sized grid) instead of sharding the FIXED "
                         "workload (the default, BASELINE config 4; \"scaling\": \"strong\")")
    ap.add_argument("--strong", action="store_true", help="accepted for compatibility: strong scaling is the default")
    ap.add_argument("--shards", default="auto", choices=["auto", "balanced", "equal"],
                    help="N > 1: equal-width contiguous shards, cost-balanced boundaries from the line positions, or (default) "
                         "balanced only where the cost model expects it to pay for the longer all-gather slot")
    ap.add_argument("--shard-of", default=None, metavar="G,r",
                    help="experiment on ONE GPU: run only shard r of a G-way sharding of the workload (no communicator): "
                         "what rank r of G would compute per step")
    ap.add_argument("--unfused", action="store_true", help="accumulate launch + separate sweep launch (A/B)")
    ap.add_argument("--step", default="auto", choices=["auto", "merged", "per-list"],
                    help="auto (default): merged, except for a cell of several line lists whose per-list accumulate launch is at most "
                         "about two rounds of workgroups (a shard of 8 of the 100-2500 cm^-1 cell: measured 0.057 per-list against 0.059 ms "
                         "merged - the merged job's 4,688 waves are just more than the chip holds at once; a shard of 4: 0.094-0.096 / 0.097). "
                         "merged: ONE accumulate job per layer over its merged, factor-weighted line lists, the layer's "
                         "absorption coefficient accumulated directly with the sweep in the kernel's output stage "
                         "(lbl_layer_merged_step_dev / lbl_layers_merged_accumulate_dev + lbl_column_fold_dev); per-list: one job and "
                         "one cross-section array per line list, then the sweep over them (lbl_layer_step_dev / lbl_column_step_dev). "
                         "Every (line, grid point) contribution is evaluated either way")
    ap.add_argument("--graph", action="store_true",
                    help="replay the captured hipGraph of the step instead of enqueueing kernel by kernel (measured slower on "
                         "ROCm 7.2: C1 0.021 vs 0.017 ms, C2 0.077 vs 0.074, a shard of 8 of C3 0.074 vs 0.071)")
    ap.add_argument("--lines", type=int, default=None, help="experiment: C2 with this many lines instead of 65,536")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--legs", default="auto",
                    help="other BASELINE configurations timed after the line's own and reported as workload_legs (untimed-region style, "
                         "not the value): a comma list of C1,C2,C3,C5 | none | auto = C1,C2,C5 beside the default workload on one GPU, "
                         "and C5 sharded N-way with its single all-gather under --gpus N")
    ap.add_argument("--no-direct-pass", action="store_true",
                    help="skip the extra untimed pass of the all-direct kernel (profiles/collect.sh: keeps the PMC passes "
                         "to the kernels of the timed path)")
    ap.add_argument("--no-api-path", action="store_true",
                    help="skip the extra legs after the timed region: the pyrad_amd.model (drop-in API) timing and, for small "
                         "cells, the steps-in-flight throughput")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--gather", default="abs_coef", choices=["abs_coef", "all"])
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-gather in stream instead of pipelined")
    ap.add_argument("--gather-batch", default="auto",
                    help="N > 1: steps of a resident set whose shards leave in ONE all-gather (fewer, larger collectives); auto = 1: "
                         "a collective per step; fit: 3..8 by the step count; or a number")
    ap.add_argument("--in-flight", default="auto", choices=["auto", "1", "2", "3"],
                    help="independent steps in flight per rank, each on a HIP stream (context) of its own; auto: 1 "
                         "(see steps_in_flight)")
    ap.add_argument("--accuracy", default="exact", choices=["exact", "budget"],
                    help="exact (default): every array within 1e-14 of the reference's fp64 values; budget: <= 1e-9 relative on the "
                         "absorption coefficient (north_star asks 1e-6), still fp64 (lbl_set_option accuracy).  The exact line "
                         "carries the budget mode's step as an extra untimed leg, `budget_leg`")
    ap.add_argument("--blocks", type=int, default=5,
                    help="the steady state is timed in this many blocks of --steps steps: block 0 is the timed region "
                         "(ms_per_step, value); the others only feed ms_per_step_blocks (median, spread)")
    ap.add_argument("--check", action="store_true", help="also compare one shard against the oracle (slow)")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="experiment: lbl_set_option(KEY, VALUE) on every context (e.g. accum_skew=0)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    # No launcher environment and N > 1: this process only starts the N ranks (it never touches the
    # GPU, so nothing that has initialised HIP is forked or replaced) and relays rank 0's line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(run_ranks(spawn_plan(args.gpus, sys.argv[1:], os.environ)))

    from pyrad_amd import _native as nat, engine, dist
    rank, local_rank, world = dist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; use "
                         "--nproc-per-node %d (or no launcher: bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    ndev = nat.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    if local_rank >= ndev:
        raise SystemExit("bench.py: rank %d (LOCAL_RANK %d) has no GPU of its own: %d device(s) visible, one process "
                         "per GPU is required" % (rank, local_rank, ndev))
    strong = not args.weak
    cfg, desc = build_workload(args.workload, (1 if strong else world) * args.scale)
    if world > 1:
        desc += (" [fixed workload, grid sharded x%d]" % world) if strong else \
                (" [grid grows with N: x%d]" % world)
    if args.scale != 1:
        desc += " [--scale %d: not a BASELINE configuration]" % args.scale
    if args.lines is not None and args.workload == "C2":
        from pyrad_amd import synthetic as _syn
        cfg = _syn.config_c2(n_lines=args.lines, range_min=500, range_max=500 + 400 * world * args.scale, seed=2)
        desc += " [--lines %d: not a BASELINE configuration]" % args.lines
    if args.workload == "C5":
        layer_cfgs = [dict(c, molecules=molecules_of(c)) for c in cfg["layers"]]
    else:
        mols = molecules_of(cfg)
        layer_cfgs = [dict(cfg, molecules=mols)]
    shard = None
    shard_world, shard_rank = world, rank
    if args.shard_of:
        shard_world, shard_rank = (int(v) for v in args.shard_of.split(","))
        desc += " [ONLY shard %d of %d: a one-GPU experiment, not a BASELINE configuration]" % (shard_rank, shard_world)
    shard_choice = "none"
    if shard_world > 1:
        shard, shard_choice = engine.choose_shards(layer_cfgs, shard_world, shard_rank, args.shards)

    # one context (HIP stream + scratch arenas) per step in flight
    g0 = engine.layer_grid(layer_cfgs[0]["P"], layer_cfgs[0]["range_min"], layer_cfgs[0]["range_max"],
                           layer_cfgs[0]["base_resolution"], layer_cfgs[0].get("dynamic_resolution", True))
    n_lists = sum(len(m["isotopologues"]) for c in layer_cfgs for m in c["molecules"])
    n_flight = steps_in_flight(args.in_flight, shard_world > 1)
    small_cell = partial_round(float(g0["n_work"]) * n_lists / shard_world)
    ctx = _Contexts([nat.Context(local_rank) for _ in range(n_flight)])
    info = ctx.first.device_info()
    if args.variant is not None:
        ctx.set_option("accum_variant", args.variant)
    if args.points_per_lane is not None:
        ctx.set_option("accum_points_per_lane", args.points_per_lane)
    if args.line_split is not None:
        ctx.set_option("accum_line_split", args.line_split)
    if args.tile_order is not None:
        ctx.set_option("accum_tile_order", args.tile_order)
    if args.blocks_per_cu is not None:
        ctx.set_option("accum_blocks_per_cu", args.blocks_per_cu)
    if args.longest_first is not None:
        ctx.set_option("accum_longest_first", args.longest_first)
    for kv in args.set:
        key, _, val = kv.partition("=")
        ctx.set_option(key, int(val))          # ("debug_*" keys exist in diagnostic builds of the library only)
    if args.accuracy == "budget":
        ctx.set_option("accuracy", 1)
    ablated = any(kv.partition("=")[0].startswith("debug_") and int(kv.partition("=")[2]) != 0 for kv in args.set)

    comm = None
    rdzv = None
    force_comm = os.environ.get("PYRAD_FORCE_COMM") == "1"      # exercise the RCCL path with one rank
    if world > 1 or force_comm:
        rdzv = dist.FileRendezvous(rank, world)
        with _StdoutToStderr():
            uid = rdzv.broadcast("rccl_unique_id", nat.Comm.unique_id() if rank == 0 else None)
            comm = nat.Comm(ctx.first, uid, world, rank)      # ONE communicator: it orders every collective against the context that owns the buffers

    t_setup = time.perf_counter()
    # Resident sets: one per step in flight (set s on context s).  With a communicator and one step in
    # flight the steps are still software-pipelined over two buffer sets of the one context: the all-gather
    # of step k (communicator stream) overlaps the kernels of step k+1 (context stream, other set).
    n_sets = n_flight if n_flight > 1 else (2 if (comm is not None and not args.no_overlap) else 1)
    overlap_gather = comm is not None and not args.no_overlap
    set_ctx = [ctx.all[i % n_flight] for i in range(n_sets)]
    if args.workload == "C5":
        layers = [engine.ResidentColumn(c, layer_cfgs, cfg["surface_T"], shard=shard) for c in set_ctx]
    else:
        layers = [engine.ResidentLayer(c, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"],
                                       mols, cfg["base_resolution"], cfg.get("dynamic_resolution", True),
                                       shard=shard) for c in set_ctx]
    layer = layers[0]
    is_column = args.workload == "C5"
    n_batch = gather_batch(args.gather_batch, overlap_gather and (is_column or args.gather == "abs_coef"), args.steps // n_sets)
    batches = ([_Batch(set_ctx[i], comm, rank, world, layers[i].S, n_batch, (2 * i, 2 * i + 1)) for i in range(n_sets)]
               if n_batch > 1 else None)
    # setup also builds the host-side schedule of every resident set (dispatch order + per-span line
    # ranges, cached by the library per line lists and grid): one priming pass each, outside the
    # timed region whatever --warmup is
    # (auto: per-list for a cell of several line lists whose per-list accumulate launch is at most about two rounds of workgroups:
    #  measured on shards of the 100-2500 cm^-1 cell, per-list / merged: of 8 0.0575 / 0.0589 ms, of 4 0.0943-0.0961 / 0.0970-0.0975,
    #  of 2 0.1658-0.1680 / 0.1559-0.1577, the whole cell 0.306 / 0.260)
    merged = choose_step(args.step, g0["n_work"], n_lists, len(layer_cfgs), shard_world, args.unfused, args.variant)
    step_kwargs = (dict(layer_arrays=bool(args.column_layer_arrays), merged=merged) if args.workload == "C5"
                   else dict(surface_T=288.0, fused=not args.unfused, merged=merged))

    def prime(L):
        L.enqueue(**step_kwargs)

    for L in layers:
        prime(L)
    ctx.sync()
    # --graph: the step of every resident set captured once (same kernels, same arguments); a step is then
    # ONE graph launch on the host side.  Steps whose kernels carry timing events are enqueued kernel by
    # kernel (events cannot sit inside the graph); both routes run the same kernels.
    graphs = [L.capture_step(**step_kwargs) for L in layers] if args.graph else [None] * len(layers)
    ctx.sync()
    # ... and brings the GPU to its sustained clocks: the first few hundred steps after an idle period
    # run up to 10 % slower (C2: 0.083 ms/step over the first 50 steps, 0.0755 after 800), which a
    # short --warmup does not cover.  Time-based, not counted as warm-up steps, outside the timed region.
    t_cond = time.perf_counter()
    while time.perf_counter() - t_cond < args.precondition_seconds:
        for _ in range(20):
            prime(layers[0])
        ctx.sync()
    t_setup = time.perf_counter() - t_setup

    # small device buffers for the RCCL barrier / max-over-ranks reduction
    red = ctx.first.buffer(max(world, 1))

    def gather_bufs(L):
        return (L.abs_coef,) if args.gather == "abs_coef" else (L.abs_coef, L.trans, L.I_out)

    def barrier():
        if comm is not None:
            if batches:
                for bt in batches:
                    bt.flush()          # a partly filled batch leaves too (every rank is at the same step)
            comm.fence_dev(-1)          # every context's stream waits for its own outstanding gathers
            ctx.sync()                  # ... and this rank's kernels and gathers are complete
            comm.allgather_dev(red, rank, 1, red)        # cross-rank barrier
        ctx.sync()

    step_no = [0]

    def step(timed_kernels=False):
        k = step_no[0]
        step_no[0] += 1
        L = layers[k % n_sets]
        bt = batches[k % n_sets] if batches else None
        if bt is not None:
            bt.before_step()
        elif overlap_gather and n_sets > 1:
            comm.fence_dev(k % n_sets)          # the gather that last used this set (step k - n_sets) is done
        slot = (k % n_sets) if (overlap_gather and n_sets > 1) else None
        g = graphs[k % n_sets]
        if g is not None and not timed_kernels:
            g.launch()
        else:
            L.enqueue(**step_kwargs)
        if bt is not None:
            bt.stage(L.I_toa if is_column else L.abs_coef, L.send_range()[0])      # the batch's B-th shard sends it
        elif comm is not None:
            if is_column:
                L.enqueue_allgather(comm, overlap_slot=slot)
            else:
                L.enqueue_allgather(comm, gather_bufs(L), overlap_slot=slot)

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events bracket only the dominant kernel inside the timed region, and only in every
    # --profile-every-th step: a pair of event records costs the stream ~5 us, 12 % of a C2 step if
    # every launch carries one.  The other kernel classes are timed in a short untimed pass afterwards.
    ctx.profile_enable(False)
    ctx.profile_reset()
    ctx.profile_reserve(2 * 8 * (args.steps // max(1, args.profile_every) + 1))      # no event creation in the timed region
    every = max(1, args.profile_every)
    n_sampled = 0
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        sampled = k % every == 0
        if sampled or (k % every == 1):
            ctx.profile_enable(["xsec_accumulate"] if sampled else False)
        n_sampled += sampled
        step(timed_kernels=sampled)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    prof = ctx.profile_read()
    # Steady state once more in blocks of `steps` (no timing events inside, barrier on both sides of each): the timed
    # region above is the driver-checked figure, these say how much a block of that length scatters on this box
    block_ms = [elapsed / args.steps * 1e3]
    for _ in range(max(0, args.blocks - 1)):
        barrier()
        t_b0 = time.perf_counter()
        for k in range(args.steps):
            step()
        barrier()
        block_ms.append((time.perf_counter() - t_b0) / args.steps * 1e3)
    ctx.profile_enable(["line_prep", "regrid", "layer_sweep", "column_sweep"] + ([] if overlap_gather else ["allgather"]))
    ctx.profile_reset()
    n_extra = max(2, min(5, args.steps))
    for _ in range(n_extra):
        step(timed_kernels=True)
    barrier()
    extra = ctx.profile_read()
    for name in ("line_prep", "regrid", "layer_sweep", "column_sweep", "allgather"):
        n_, ms_ = extra[name]
        prof[name] = (n_ * args.steps // n_extra, ms_ * args.steps / n_extra)      # scaled to the timed step count
    # the all-direct kernel (variant 3: every (line, grid point) pair evaluated, no series) on the
    # same resident inputs, untimed, so that the line carries both numbers
    direct_ms = None
    if args.variant in (None, 5) and not args.no_direct_pass:
        ctx.set_option("accum_variant", 3)
        for _ in range(2):
            step(timed_kernels=True)
        barrier()
        ctx.profile_enable(["xsec_accumulate"])
        ctx.profile_reset()
        for _ in range(n_extra):
            step(timed_kernels=True)
        barrier()
        n_d, ms_d = ctx.profile_read()["xsec_accumulate"]
        direct_ms = ms_d / n_extra
        ctx.set_option("accum_variant", 5)
    ctx.profile_enable(False)
    # the same step in budget mode (<= 1e-9 relative on the absorption coefficient), untimed legs: wall clock over
    # n_extra x 4 steps between barriers, then every kernel class bracketed by events in a pass of its own
    budget_leg = None
    if args.accuracy == "exact" and args.variant in (None, 5) and not args.no_direct_pass:
        ctx.set_option("accuracy", 1)
        for _ in range(3):
            step()
        barrier()
        n_bl = max(10, min(40, args.steps))
        t_b0 = time.perf_counter()
        for _ in range(n_bl):
            step()
        barrier()
        t_bl = (time.perf_counter() - t_b0) / n_bl
        ctx.profile_enable(["line_prep", "xsec_accumulate", "layer_sweep", "column_sweep"])
        ctx.profile_reset()
        for _ in range(n_extra):
            step(timed_kernels=True)
        barrier()
        pb = ctx.profile_read()
        ctx.profile_enable(False)
        ctx.set_option("accuracy", 0)
        for _ in range(2):
            step()
        barrier()
        budget_leg = {"ms_per_step": t_bl * 1e3, "evals_per_s": float(layer.evals) / t_bl,
                      "kernel_ms_per_step": {k_: pb[k_][1] / n_extra for k_ in ("line_prep", "xsec_accumulate", "layer_sweep", "column_sweep")},
                      "what": "the same resident step with lbl_set_option accuracy = 1 (budget: 18..7 far-field series terms by distance, Gaussian cut-off "
                              "2^-34; tests hold it to 1e-9 relative on the absorption coefficient at every grid point): %d steps between barriers, then %d steps with every "
                              "kernel class bracketed by events; not the line's value" % (n_bl, n_extra)}

    # the per-line-list step (one job and one cross-section array per line list, then the sweep over them) on the same
    # resident inputs, untimed legs like the budget leg's: the line's value is the merged step, this is what it replaced
    per_list_leg = None
    other_ok = merged or (args.step == "auto" and not args.unfused)          # (auto chose per-list: the merged step is the leg)
    if other_ok and args.variant in (None, 5) and not args.no_direct_pass:
        step_kwargs["merged"] = not merged
        for _ in range(3):
            step()
        barrier()
        n_pl = max(10, min(40, args.steps))
        t_p0 = time.perf_counter()
        for _ in range(n_pl):
            step()
        barrier()
        t_pl = (time.perf_counter() - t_p0) / n_pl
        ctx.profile_enable(["line_prep", "xsec_accumulate", "layer_sweep", "column_sweep"])
        ctx.profile_reset()
        for _ in range(n_extra):
            step(timed_kernels=True)
        barrier()
        pp = ctx.profile_read()
        ctx.profile_enable(False)
        step_kwargs["merged"] = merged
        for _ in range(2):
            step()
        barrier()
        per_list_leg = {"step": "per-list" if merged else "merged", "ms_per_step": t_pl * 1e3, "evals_per_s": float(layer.evals) / t_pl,
                        "kernel_ms_per_step": {k_: pp[k_][1] / n_extra for k_ in ("line_prep", "xsec_accumulate", "layer_sweep", "column_sweep")},
                        "xsec_accumulate_launches_per_step": pp["xsec_accumulate"][0] / n_extra,
                        "what": ("the same resident inputs through the per-line-list step (--step per-list: lbl_layer_step_dev / "
                                 "lbl_xsec_accumulate_dev + lbl_column_step_dev - one accumulate job and one cross-section array per line "
                                 "list, then the sweep kernel over them; the step of rounds 1-4)" if merged else
                                 "the same resident inputs through the merged step (--step merged: one accumulate job per layer over its "
                                 "merged, factor-weighted line lists, sweep in its output stage)") +
                                ": %d steps between barriers, then %d steps with every kernel class bracketed by events; same "
                                "evals_per_step; not the line's value" % (n_pl, n_extra)}

    # With a communicator: where the sharded step's time goes, in two short untimed passes (every rank runs
    # them in lockstep): the kernels of a step without the all-gather, and the all-gather alone, in stream.
    # The pipelined step of the timed region cannot be faster than the longer of the two.
    breakdown_local = None
    if comm is not None:
        n_b = max(5, min(20, args.steps))
        if batches:
            n_b = max(n_batch, n_b // n_batch * n_batch)       # whole batches
        barrier()
        t_b = time.perf_counter()
        for i_b in range(n_b):
            layers[i_b % n_sets].enqueue(**step_kwargs)
        barrier()
        t_compute = (time.perf_counter() - t_b) / n_b
        t_b = time.perf_counter()
        if batches:
            for _ in range(n_b // n_batch):
                batches[0].send(overlap=False)                  # the collective of one batch, in stream
        else:
            for _ in range(n_b):
                if is_column:
                    layers[0].enqueue_allgather(comm)
                else:
                    layers[0].enqueue_allgather(comm, gather_bufs(layers[0]))
        barrier()
        t_gather = (time.perf_counter() - t_b) / n_b

        # Like-for-like legs, whatever mode the timed region ran in: ONE step in flight on ONE resident set, ONE
        # collective per step; first with the all-gather in stream (a step's spectrum is complete before the next
        # step starts), then overlapped with the next step's kernels on the communicator's stream (two buffer
        # sets when the run has them, else the same set: a slot's fence orders the reuse).
        def plain_step(L, overlap_slot=None):
            L.enqueue(**step_kwargs)
            if is_column:
                L.enqueue_allgather(comm, overlap_slot=overlap_slot)
            else:
                L.enqueue_allgather(comm, gather_bufs(L), overlap_slot=overlap_slot)

        barrier()
        t_b = time.perf_counter()
        for _ in range(n_b):
            plain_step(layers[0])
        barrier()
        t_instream = (time.perf_counter() - t_b) / n_b
        t_overlap = float("nan")
        if n_sets >= 2:                                         # (one buffer set: nothing to overlap a gather with)
            barrier()
            t_b = time.perf_counter()
            for i_b in range(n_b):
                sl = i_b % 2
                comm.fence_dev(4 + sl)                          # the gather that used this set two steps ago
                plain_step(layers[sl], overlap_slot=4 + sl)
            barrier()
            t_overlap = (time.perf_counter() - t_b) / n_b
        # latency of ONE step: enqueue -> the gathered spectrum usable on the host side (stream drained), median of 7
        lat = []
        for _ in range(7):
            barrier()
            t_b = time.perf_counter()
            plain_step(layers[0])
            ctx.sync()
            lat.append(time.perf_counter() - t_b)
        t_latency = sorted(lat)[len(lat) // 2]
        breakdown_local = (t_compute, t_gather, t_instream, t_overlap, t_latency)

    # Batched gather: one more full batch per resident set, then every slot of the gathered buffer is checked on
    # the host - this rank's own slots against its shard bit for bit, every rank's slots finite, positive over
    # the rank's points (non-negative, not all zero) and identical from step to step (the steps repeat the same cell).  A mismatch is fatal.
    gather_verified = None
    if batches:
        def shard_src(L):
            return L.I_toa if is_column else L.abs_coef
        for i, bt in enumerate(batches):
            for _ in range(n_batch):
                bt.before_step()
                layers[i].enqueue(**step_kwargs)
                bt.stage(shard_src(layers[i]), layers[i].send_range()[0])
        barrier()
        for i, bt in enumerate(batches):
            L = layers[i]
            S_ = L.S
            a = bt.bufs[bt.cur ^ 1].download(world * n_batch * S_).reshape(world, n_batch, S_)
            own = shard_src(L).download(S_, L.send_range()[0])
            counts = [c for _, c in L.plan.bounds] if (L.plan is not None and L.plan.world == world) else [L.count] * world
            good = bool(np.all(np.isfinite(a)))
            for b in range(n_batch):
                good = good and np.array_equal(a[rank, b], own)
                for r in range(world):
                    good = good and np.array_equal(a[r, b], a[r, 0])
            for r in range(world):
                x = a[r, 0, :counts[r]]
                good = good and (x.size == 0 or bool(np.all(x >= 0.0) and np.any(x > 0.0)))
            if not good:
                raise SystemExit("bench.py: rank %d: the batched all-gather of resident set %d did not deliver the shards" % (rank, i))
        gather_verified = True

    # max over ranks of the elapsed time, sum over ranks of the evals — through the one comm
    evals_local = float(layer.evals)
    if comm is not None:
        red.upload(np.array([elapsed], dtype=np.float64), offset=rank)
        comm.allgather_dev(red, rank, 1, red)
        times = red.download(world)
        red.upload(np.array([evals_local], dtype=np.float64), offset=rank)
        comm.allgather_dev(red, rank, 1, red)
        evals_all = red.download(world)
        elapsed_max = float(times.max())
        evals_total = float(evals_all.sum())
        per_rank = []
        for v in breakdown_local:
            red.upload(np.array([v], dtype=np.float64), offset=rank)
            comm.allgather_dev(red, rank, 1, red)
            per_rank.append(red.download(world).copy())
        n_gathered = 1 if (is_column or args.gather == "abs_coef") else 3
        recv_bytes = (world - 1) * layer.S * 8.0 * n_gathered
        t_ag = float(per_rank[1].max())
        breakdown = {"kernels_only_ms_per_step": float(per_rank[0].max()) * 1e3,
                     "kernels_only_ms_by_rank": [round(float(v) * 1e3, 4) for v in per_rank[0]],
                     "allgather_alone_ms_per_step": t_ag * 1e3,
                     "allgather_alone_GBps_per_rank": (recv_bytes / t_ag / 1e9) if t_ag > 0 else None,
                     "allgather_bytes_received_per_rank_per_step": recv_bytes,
                     "like_for_like": {
                         "in_stream_ms_per_step": float(per_rank[2].max()) * 1e3,
                         "overlapped_ms_per_step": (float(per_rank[3].max()) * 1e3) if n_sets >= 2 else None,
                         "step_latency_ms": float(per_rank[4].max()) * 1e3,
                         "what": "ONE step in flight, ONE collective per step, whatever mode the timed region ran in: the "
                                 "all-gather in stream after the step's kernels; the same with the gather overlapped with the "
                                 "next step's kernels (communicator stream); and the latency of a single step from enqueue to "
                                 "the gathered spectrum (stream drained, median of 7); wall clock between barriers, max over ranks. "
                                 "Form N-GPU / 1-GPU ratios of this mode with the N = 1 line's ms_per_step (one step in flight)"},
                     "what": "two short untimed passes after the timed region, wall clock between barriers, max over "
                             "ranks: the step's kernels with no all-gather (%d step(s) in flight, as in the timed region), and "
                             "the step's all-gather(s) alone in stream (%s%s); the pipelined step overlaps the two"
                             % (n_flight, "outgoing spectrum" if is_column else args.gather,
                                "" if not batches else ", one collective per %d steps, per step" % n_batch)}
    else:
        elapsed_max, evals_total = elapsed, evals_local
        breakdown = None

    result = None
    if rank == 0:
        world_scale = evals_total / max(evals_local, 1.0)       # (rank 0's pair counts scaled to the job: shards are alike)
        value = evals_total * args.steps / elapsed_max
        n_acc, ms_acc = prof["xsec_accumulate"]
        n_sw, ms_sw = prof["layer_sweep"]
        n_prep, ms_prep = prof["line_prep"]
        n_ag, ms_ag = prof["allgather"]
        g = layer.layers[0].g if is_column else layer.g
        pts = layer.count if layer.plan is not None else g["n_work"]
        n_arrays = len(layer.layers[0].jobs) if is_column else len(layer.jobs)     # cross-section arrays per layer
        # Algorithmic bytes per launch of the dominant kernel: every line's 7 fp64 HITRAN
        # fields read once (56 B/line) + 8 B per grid point for EVERY array the launch writes: one
        # cross section per line list, and k, transmittance, outgoing radiance when the sweep is fused in.
        fused_sweep = (not is_column) and not args.unfused and n_sw == 0
        t_acc = (ms_acc / max(n_acc, 1)) * 1e-3
        t_acc_step = (ms_acc / max(n_sampled, 1)) * 1e-3        # all K2 launches of one (sampled) step
        launches_per_step = max(n_acc // max(n_sampled, 1), 1)   # a column launches K2 once per window group
        n_xsec_written = (n_arrays * len(layer.layers)) if is_column else n_arrays
        if merged:          # no per-line-list cross sections: a column writes one absorption coefficient per layer, a cell none beside the sweep's
            n_xsec_written = len(layer.layers) if is_column else 0
        balg_acc = (56.0 * layer.n_lines + 8.0 * pts * n_xsec_written + (24.0 * pts if fused_sweep else 0.0)) / launches_per_step
        achieved = balg_acc / t_acc / 1e9 if t_acc > 0 else 0.0
        # per sweep launch: M xsec reads + k, T (, I_out) writes; I_in is computed in-kernel
        balg_sw = 8.0 * pts * (n_arrays + 3)
        sweep_kernel = "layer_sweep_kernel"
        if is_column:
            # one column_step_kernel launch per step: every layer's cross sections read once, the outgoing
            # spectrum written (+ two arrays per layer when --column-layer-arrays 1)
            n_sw, ms_sw = prof["column_sweep"]
            n_layers = len(layer.layers)
            balg_sw = 8.0 * pts * (n_arrays * n_layers + 1 + (2 * n_layers if args.column_layer_arrays else 0))
            if merged:      # one absorption coefficient per layer read, the outgoing spectrum written (+ the layers' transmittances)
                balg_sw = 8.0 * pts * (n_layers + 1 + (n_layers if args.column_layer_arrays else 0))
            sweep_kernel = "column_step_kernel"
        t_sw = (ms_sw / max(n_sw, 1)) * 1e-3
        # committed PMC passes: of the whole cell, or - for a rank of a G-way sharded run - of one shard of G
        # ("C3s8": shard 4 of 8, profiles/collect.sh r02 C3 8,4): a shard's launch shape, not this rank's exact lines
        standard = args.scale == 1 and args.lines is None and not args.unfused and (strong or world == 1)
        pmc_key = args.workload if shard_world == 1 else "%ss%d" % (args.workload, shard_world)
        pmc = load_pmc(pmc_key if standard else None, nat.source_hash())
        traffic = pmc["hbm"].get("xsec_accumulate_kernel")
        result = {
            "metric": "line*gridpoint evals/sec (whole job)", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "grid_points_per_gpu": int(pts), "lines_per_gpu": int(layer.n_lines),
                       "window_W": int(g["W"]), "evals_per_step": evals_total, "parallelism": "grid-range x%d" % world,
                       "step": ("merged: ONE accumulate job per layer over its merged, factor-weighted line lists - every line's amplitude "
                                "times its molecule's conc P / 1E4 / k / T, all records of the layer in one centre-index-ordered array - so that "
                                "the kernel accumulates the layer's absorption coefficient sum_m f_m sum_iso xs_iso (pyradClasses.py:707-712, "
                                "581-583, 566-571) directly, with transmittance and radiance in its output stage (a column: the fold over the "
                                "layers' absorption coefficients); every (line, grid point) contribution is evaluated (evals_per_step unchanged); "
                                "no per-line-list cross-section array is written - lbl_xsec_accumulate_dev produces one on demand, as the "
                                "reference's lazy getters do (cls:32-88); the per-list step is the per_list_leg"
                                if merged else
                                "per-list: one accumulate job and one cross-section array per line list, then the sweep over them" +
                                ("" if args.step != "auto" else " (--step auto: this cell's per-list accumulate launch is at most about two rounds of "
                                 "workgroups, where the merged job measured no faster; the merged step is the merged_leg)")),
                       "accuracy": args.accuracy, "gathered": args.gather, "device": info["name"], "preconditioning_s": args.precondition_seconds,
                       "shard_bounds": (None if layer.plan is None else [list(b) for b in layer.plan.bounds]),
                       "shards": shard_choice,
                       "step_launch": ("" if n_flight == 1 else "%d independent steps in flight, each on a HIP stream of its "
                                       "own (ms_per_step is the timed region / steps, not a step's latency); " % n_flight) +
                                      ("kernel by kernel" if not args.graph else
                                       "one hipGraph per step (K1, K2, sweep captured once); kernel by kernel in the steps "
                                       "that carry timing events"),
                       "steps_in_flight": n_flight,
                       "gather_batch": n_batch, "gather_verified": gather_verified,
                       "allgather": ("none" if comm is None else "in-stream" if not overlap_gather else
                                     "overlapped with the next step (%d buffer sets)" % n_sets if not batches else
                                     "one collective per %d steps of a resident set (shards staged side by side in a batch "
                                     "buffer), overlapped with the steps that fill the set's other batch buffer" % n_batch)},
            "pairs": dict(layer.pairs, **{
                "what": "how the default kernel treats the step's (line, span of 256 grid points) pairs: pairs_series go through "
                        "the far-field series about the span centre (30 terms at 4 half-spans, 20 / 15 / 12 from 8 / 16 / 32 on: the remainder stays below half an ulp; budget mode 18 / 12 / 9 / 7), pairs_direct "
                        "are evaluated point by point; evals_series = 256 per far pair, evals_direct the rest of evals_per_step. "
                        "`value` counts every contribution the reference's loop adds (both kinds); value_direct_kernel is the "
                        "same workload through the all-direct kernel (accum_variant 3), every pair evaluated point by point"}),
            "evals_direct_per_s": float(layer.pairs["evals_direct"]) * world_scale * args.steps / elapsed_max,
            "value_direct_kernel": None,
            "direct_frac": None,
            "roofline": {"bound": "hbm", "kernel": "xsec_accumulate_lds_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_stale": (pmc["stale"] if traffic is not None else None),
                         "traffic_source": pmc["source"],
                         "algorithmic_bytes_per_launch": balg_acc, "avg_launch_ms": t_acc * 1e3, "launches": n_acc,
                         "launches_timed": "every launch of every %d-th timed step (%d of %d steps)" % (every, n_sampled, args.steps),
                         "sweep_fused_in": bool(fused_sweep),
                         "launches_overlap": n_flight > 1,
                         "note": "compulsory traffic only (56 B/line + 8 B/grid point per array written: one cross section "
                                 "per line list - none in the merged step, one absorption coefficient per layer in a merged column - "
                                 "+ k, transmittance and radiance when the layer sweep is fused in). This "
                                 "kernel is fp64-VALU bound by construction (SURVEY.md §8d): valu_f64.busy_frac is its "
                                 "real utilisation figure"},
            "valu_f64": valu_block(evals_local, t_acc_step, direct_ms, args.variant, pmc, launches_per_step,
                                   n_flight, elapsed_max / args.steps),
            "roofline_sweep": {"fused_into": "xsec_accumulate_lds_kernel (%s)" % ("lbl_layer_merged_step_dev" if merged else "lbl_layer_step_dev")}
                              if n_sw == 0 and not is_column else
                              {"bound": "hbm", "kernel": sweep_kernel,
                               "achieved": balg_sw / t_sw / 1e9 if t_sw > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (balg_sw / t_sw / 1e9 / HBM_PEAK_GBS) if t_sw > 0 else 0.0,
                               "traffic": pmc["hbm"].get(sweep_kernel),
                               "traffic_stale": (pmc["stale"] if pmc["hbm"].get(sweep_kernel) is not None else None),
                               "algorithmic_bytes_per_launch": balg_sw, "avg_launch_ms": t_sw * 1e3, "launches": n_sw,
                               # a pair of HIP event records costs the stream a few microseconds, which shows on a 25 us
                               # kernel: the committed rocprofv3 kernel trace of this very command is the cleaner clock
                               "avg_launch_ms_rocprofv3": (pmc["avg_us"].get(sweep_kernel) or 0.0) / 1e3 or None,
                               "frac_rocprofv3": (balg_sw / (pmc["avg_us"][sweep_kernel] * 1e-6) / 1e9 / HBM_PEAK_GBS)
                                                 if pmc["avg_us"].get(sweep_kernel) else None},
            "kernel_ms_per_step": {"line_prep": ms_prep / args.steps, "xsec_accumulate": ms_acc / max(n_sampled, 1),
                                   ("column_step" if is_column else "layer_sweep"): ms_sw / args.steps,
                                   "allgather": ms_ag / args.steps,
                                   "source": "xsec_accumulate: HIP events around its launches in every %d-th step OF THE TIMED REGION "
                                             "(%d of %d steps); line_prep, sweep%s: a SEPARATE pass of %d steps after it, every launch "
                                             "bracketed (a pair of event records costs the stream a few microseconds, so the parts "
                                             "can sum to more than ms_per_step)" % (every, n_sampled, args.steps,
                                                                                   "" if overlap_gather else ", allgather", n_extra)},
            "ms_per_step_blocks": {"blocks": [round(v, 6) for v in block_ms], "median": float(np.median(block_ms)),
                                   "min": float(np.min(block_ms)), "max": float(np.max(block_ms)),
                                   "spread_rel": float((np.max(block_ms) - np.min(block_ms)) / np.median(block_ms)),
                                   "what": "max over ranks is NOT taken here (rank 0's clock between barriers); block 0 is the "
                                           "timed region that ms_per_step and value come from"},
            "setup_s": t_setup,
        }
        vb = result["valu_f64"]
        if vb.get("direct_kernel_evals_per_s"):
            result["value_direct_kernel"] = vb["direct_kernel_evals_per_s"]
            result["direct_frac"] = vb.get("direct_frac")
        if budget_leg is not None:
            result["budget_leg"] = budget_leg
        if per_list_leg is not None:
            result["per_list_leg" if merged else "merged_leg"] = per_list_leg
        # the figure comparable with rounds 1-4 (whose step wrote one cross section per line list) stays on the top level,
        # whichever step --step auto made the line's value (advisor, round 5)
        result["value_step"] = "merged" if merged else "per-list"
        result["value_per_list"] = value if not merged else (per_list_leg["evals_per_s"] * world if per_list_leg is not None else None)
        if ablated:
            result["ablated"] = True
            result["invalid"] = "a debug_* option was set: parts of the kernels are switched off, results are wrong, timing experiment only"
        if breakdown is not None:
            result["sharded_step_breakdown"] = breakdown
        if not args.no_cpu_baseline and world == 1:
            raw = cfg["layers"] if is_column else [cfg]
            result["cpu_baseline"] = cpu_baseline([dict(c, raw_molecules=c["molecules"]) for c in raw], args.workload, args.cpu_seconds)
        if args.check:
            result["check"] = oracle_check(layer, cfg)
    want_api = rank == 0 and world == 1 and not args.no_api_path and not args.shard_of
    legs = []
    if args.legs == "auto":
        if args.workload == "C3" and not args.shard_of and args.scale == 1 and strong and args.variant is None and not args.unfused:
            legs = ["C1", "C2", "C5"] if (world == 1 and comm is None) else ["C5"]
    elif args.legs != "none":
        legs = [w for w in args.legs.split(",") if w]
    leg_results = {}
    if comm is not None and legs:
        # every rank, in lockstep: the leg's workload sharded over the same ranks through the same communicator
        for w in legs:
            barrier()
            leg_results[w] = workload_leg(w, local_rank, args.steps, args.warmup, args.step, args.accuracy, want_api=False,
                                          shards=args.shards, ctx=ctx.first, comm=comm, world=world, rank=rank, red=red)
        legs = []
    if rdzv is not None:
        rdzv.arrive("done")
        rdzv.cleanup()
    if batches:
        for bt in batches:
            bt.free()
    if comm is not None:
        comm.free()
    for g_ in graphs:
        if g_ is not None:
            g_.free()
    for L in layers:
        L.free()
    red.free()
    ctx.close()
    if want_api and is_column:
        result["api_path"] = api_path_column(cfg)
    elif want_api:
        result["api_path"] = api_path(cfg)
        if small_cell and n_flight == 1:
            result["in_flight_leg"] = in_flight_leg(cfg, merged=merged)
        elif n_flight == 1:
            result["in_flight_leg"] = in_flight_leg(cfg, n_flight=2, steps=60, merged=merged)     # so that ratios can be formed in either mode
    if rank == 0 and world == 1:
        for w in legs:
            leg_results[w] = workload_leg(w, local_rank, args.steps, args.warmup, args.step, args.accuracy,
                                          want_api=(w == "C5" and not args.no_api_path))
    if rank == 0:
        if leg_results:
            result["workload_legs"] = leg_results
        print(json.dumps(result))


SIMDS = 256 * 4
CLOCK_HZ = 2.4e9


def valu_block(evals_local, t_acc_step, direct_ms, variant, pmc, launches_per_step, n_flight=1, t_step=None):
    """fp64 vector-ALU accounting of K2.
    busy_frac: the kernel's measured VALU utilisation, SQ_INSTS_VALU (wave-instructions per launch,
    from the committed rocprofv3 --pmc pass of this very command) x 4 issue cycles per fp64/VALU
    wave-instruction / (1024 SIMDs x the launch duration measured live in this run x 2.4 GHz).
    direct_*: the all-direct kernel (variant 3) spends at least 5 fp64 instructions per (line, grid
    point) pair; the default kernel (variant 5) replaces the pairs of distant Lorentz lines by a
    30-term series per (line, span), so its PAIR rate is not bounded by 5 instructions per pair and
    only the direct kernel's rate is priced per pair against the VALU peak."""
    far_field = variant in (None, 5)
    out = {"kernel_evals_per_s": evals_local / t_acc_step if t_acc_step > 0 else 0.0,
           "far_field_series": far_field, "instr_per_eval_direct": FP64_INSTR_PER_EVAL,
           "peak_lane_instr_per_s": FP64_VALU_PEAK_INSTR}
    insts = pmc["valu"].get("xsec_accumulate_kernel" if far_field else "xsec_accumulate_direct_kernel")
    if insts and t_acc_step > 0 and n_flight > 1 and t_step:
        # launches of different steps overlap: a launch's own duration no longer says how busy the chip is.
        # K2's VALU instructions of one step over the step time of the timed region (K1 / sweep not counted)
        out.update({"busy_frac": insts * launches_per_step * 4.0 / (SIMDS * t_step * CLOCK_HZ),
                    "valu_wave_insts_per_launch": insts, "busy_frac_stale": pmc["stale"],
                    "busy_frac_source": "%s SQ_INSTS_VALU x %d launch(es) per step over the timed region's %.1f us per step "
                                        "(%d steps in flight: kernel durations overlap)" % (
                                            pmc["source"], launches_per_step, t_step * 1e6, n_flight)})
    elif insts and t_acc_step > 0:
        t_launch = t_acc_step / launches_per_step
        out.update({"busy_frac": insts * 4.0 / (SIMDS * t_launch * CLOCK_HZ),
                    "valu_wave_insts_per_launch": insts, "busy_frac_stale": pmc["stale"],
                    "busy_frac_source": "%s SQ_INSTS_VALU, live launch time %.1f us (rocprofv3 average of that pass: %s us)" % (
                        pmc["source"], t_launch * 1e6,
                        ("%.1f" % pmc["avg_us"]["xsec_accumulate_kernel"]) if "xsec_accumulate_kernel" in pmc["avg_us"] else "n/a")})
    else:
        out.update({"busy_frac": None, "busy_frac_stale": None})
    t_direct = direct_ms * 1e-3 if direct_ms else (None if far_field else t_acc_step)
    if t_direct:
        rate = evals_local / t_direct
        out.update({"direct_kernel_ms_per_step": t_direct * 1e3, "direct_kernel_evals_per_s": rate,
                    "direct_achieved_lane_instr_per_s": FP64_INSTR_PER_EVAL * rate,
                    "direct_frac": FP64_INSTR_PER_EVAL * rate / FP64_VALU_PEAK_INSTR})
    out["note"] = ("busy_frac = measured VALU issue utilisation of the default kernel; 5 fp64 instr per eval is the "
                   "running-fraction Lorentz loop's minimum (measured ceiling of that loop alone on this chip: 4.9e12 "
                   "evals/s, scripts/ubench_fp64.hip); direct_* = the all-direct kernel (accum_variant 3) on the same "
                   "inputs, timed in an extra untimed pass")
    return out


def load_pmc(workload, source_hash):
    """Committed rocprofv3 --pmc results of this workload (profiles/pmc_traffic.json, produced by
    profiles/collect.sh + summarize.py on the GPU box in separate FETCH_SIZE / WRITE_SIZE / SQ passes;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be read
    inside a timed run, so these are the values measured for the same command; `stale` says whether
    the kernel sources have changed since (hash of csrc/ + the header); no entry -> nulls."""
    out = {"hbm": {}, "valu": {}, "avg_us": {}, "stale": None, "source": None}
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if workload is None or not os.path.isfile(path):
        return out
    try:
        with open(path) as f:
            e = json.load(f).get(workload, {})
    except (OSError, ValueError):
        return out
    if not e:
        return out
    out["hbm"] = e.get("hbm_bytes_per_launch", {})
    out["valu"] = e.get("valu_wave_insts_per_launch", {})
    out["avg_us"] = e.get("rocprofv3_avg_us", {})
    out["stale"] = e.get("source_hash") != source_hash
    out["source"] = "profiles/%s" % e.get("source")
    return out


def oracle_check(layer, cfg):
    from oracle import pyrad_oracle as orc
    ref = orc.layer_properties(cfg)
    got = layer.results()
    first, count = (layer.first, layer.count) if layer.plan is not None else (0, layer.n)
    sl = slice(first, first + count)
    a, b = got["abs_coef"][sl], ref["abs_coef"][sl]
    return {"max_rel_err_abs_coef": float(np.max(np.abs(a - b) / np.abs(b)))}


if __name__ == "__main__":
    main()
