"""Diagnostic (GPU): latency of the one-shot host-in / host-out entry point (lbl_xsec_accumulate through
the ctypes binding) on C1 and C2, with and without the host schedule: the PCIe-inclusive rate of DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pyrad_amd import _native as nat, engine, synthetic
ctx = nat.Context(0)
for name, cfg in (("C1", synthetic.config_c1()), ("C2", synthetic.config_c2())):
    lines = cfg["molecules"][0]["lines"]
    g = engine.layer_grid(cfg["P"], cfg["range_min"], cfg["range_max"], cfg["base_resolution"], cfg["dynamic_resolution"])
    sel = engine.select_window(lines, g["eff_min"], g["eff_max"])
    sp = synthetic.SPECIES["co2"]
    iso = nat.IsoParams(296.0, 1013.25, 4e-4, sp["molmass"], synthetic.q_value("co2", 296), sp["q296"])
    G = engine.native_grid(g)
    for lf in (3, 0):
        ctx.set_option("accum_longest_first", lf)
        ctx.xsec_accumulate(sel, iso, G)
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.xsec_accumulate(sel, iso, G)
        print(name, "longest_first", lf, "one-shot ms per call %.3f" % ((time.perf_counter() - t0) / 10 * 1e3))
