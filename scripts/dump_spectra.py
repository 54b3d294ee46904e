#!/usr/bin/env python3
"""Absorption coefficients of a few cells through the resident path, saved as .npy (one file per cell): run once per
library build (PYRAD_HIP_LIB) and compare the files bit for bit.  usage: dump_spectra.py <out dir>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine
import bench
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
ctx = nat.Context(0)
for w, merged in (("C3", True), ("C3", False), ("C2", True), ("C1", True)):
    cfg, _ = bench.build_workload(w, 1)
    L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                             cfg["base_resolution"], cfg.get("dynamic_resolution", True))
    L.enqueue(surface_T=288.0, merged=merged)
    ctx.sync()
    r = L.results()
    np.save(os.path.join(out, "%s_%s_k.npy" % (w, "merged" if merged else "perlist")), r["abs_coef"])
    np.save(os.path.join(out, "%s_%s_I.npy" % (w, "merged" if merged else "perlist")), r["transmission"])
    L.free()
cfg, _ = bench.build_workload("C5", 1)
layer_cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
col = engine.ResidentColumn(ctx, layer_cfgs, cfg["surface_T"])
col.enqueue(layer_arrays=True, merged=True)
ctx.sync()
r = col.results()
np.save(os.path.join(out, "C5_toa.npy"), r["toa"])
for i in (0, 7, 12, 13, 20, 29):
    np.save(os.path.join(out, "C5_k%02d.npy" % i), col.layers[i].abs_coef.download(col.n))
col.free()
ctx.close()
print("dumped to", out)
