#!/usr/bin/env python3
"""Merged layer step (lbl_layer_merged_step_dev) against the per-line-list step (lbl_layer_step_dev) on the same
resident inputs: worst relative difference of k / transmittance / radiance, and the step times of both, for a few
shapes.  A development aid (GPU); the suite's version is tests/test_gpu_merged.py."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine, synthetic          # noqa: E402
import bench                                                      # noqa: E402


def rel(a, b):
    den = np.abs(b)
    ok = den > 0
    return float(np.max(np.abs(a[ok] - b[ok]) / den[ok])) if ok.any() else 0.0


def time_steps(ctx, fn, n=40):
    for _ in range(5):
        fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


def main():
    which = [a for a in sys.argv[1:] if "=" not in a] or ["C1", "C2", "C3"]
    ctx = nat.Context(0)
    for a in sys.argv[1:]:
        if "=" in a:
            ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
            print("option", a)
    for w in which:
        shard = None
        name = w
        if "s" in w[1:]:
            w, g = w.split("s")
            shard = (int(g), int(g) // 2)
        cfg, desc = bench.build_workload(w, 1)
        if w == "C5":
            layer_cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
            col = engine.ResidentColumn(ctx, layer_cfgs, cfg["surface_T"], shard=shard)
            col.enqueue(layer_arrays=True)
            ctx.sync()
            ref = col.results()
            ref_k = [L.abs_coef.download(col.n) for L in col.layers]
            col.enqueue(layer_arrays=True, merged=True)
            ctx.sync()
            got = col.results()
            got_k = [L.abs_coef.download(col.n) for L in col.layers]
            sl = slice(col.first, col.first + col.count)
            print("%s merged vs per-list: toa %.2e, k worst %.2e, trans worst %.2e" % (
                name, rel(got["toa"][sl], ref["toa"][sl]), max(rel(a[sl], b[sl]) for a, b in zip(got_k, ref_k)),
                max(rel(a[sl], b[sl]) for a, b in zip(got["transmittance"], ref["transmittance"]))))
            t_a = time_steps(ctx, lambda: col.enqueue(layer_arrays=False), 10)
            t_b = time_steps(ctx, lambda: col.enqueue(layer_arrays=False, merged=True), 10)
            print("%s step: per-list %.4f ms, merged %.4f ms" % (name, t_a, t_b))
            col.free()
            continue
        mols = bench.molecules_of(cfg)
        L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                 cfg["base_resolution"], cfg.get("dynamic_resolution", True), shard=shard)
        L.enqueue(surface_T=288.0)
        ctx.sync()
        ref = L.results()
        for b in (L.abs_coef, L.trans, L.I_out):
            b.fill(float("nan"))
        L.enqueue(surface_T=288.0, merged=True)
        ctx.sync()
        got = L.results()
        sl = slice(L.first, L.first + L.count)
        print("%s merged vs per-list: k %.2e trans %.2e I %.2e (n=%d lines=%d)" % (
            name, rel(got["abs_coef"][sl], ref["abs_coef"][sl]), rel(got["transmittance"][sl], ref["transmittance"][sl]),
            rel(got["transmission"][sl], ref["transmission"][sl]), L.count, L.n_lines))
        assert np.all(np.isfinite(got["abs_coef"][sl]))
        t_a = time_steps(ctx, lambda: L.enqueue(surface_T=288.0))
        t_b = time_steps(ctx, lambda: L.enqueue(surface_T=288.0, merged=True))
        print("%s step: per-list %.4f ms, merged %.4f ms" % (name, t_a, t_b))
        L.free()
    ctx.close()


if __name__ == "__main__":
    main()
