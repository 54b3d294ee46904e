#!/usr/bin/env python3
"""Where Atmosphere.transmission's time over the column step goes: the call, the same handle without the download, the
download alone, and the call with other piece counts.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyrad_amd import model, data, settings, engine
cfg, _ = bench.build_workload("C5", 1)
c0 = cfg["layers"][0]
settings.set_resolution_multiplier(c0["base_resolution"] / .01)
data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in c0["molecules"]}))
model.Layer.hasAtmosphere = False
atm = model.Atmosphere("col")
for c in cfg["layers"]:
    L = atm.addLayer(c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], name=c["name"], dynamicResolution=c.get("dynamic_resolution", True))
    for m in c["molecules"]:
        L.addMolecule(m["species"], **m["conc"])
atm.transmission(surfaceTemperature=288)
ctx = model._ctx()
def med(f, reps=7):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))
def call():
    for L in atm: L.changeTemperature(L.T)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); return time.perf_counter() - t0
for _ in range(40): call()
print("call: %.3f ms (median of 9)" % (1e3 * float(np.median([call() for _ in range(9)]))))
fast = atm.__dict__["_column_fast"]; col = fast["col"]; n = fast["glob"][1]
out = atm.__dict__["_toa_state"].bufs["toa"]
host = ctx.host_array(n)
def compute_only():
    col.transmission(None, out, host=None, surface_T=288.0, pieces=1); ctx.sync()
print("handle, all layers due, no download: %.3f ms" % med(compute_only))
for pieces in (1, 2, 4, 8, 16):
    def both():
        col.transmission(None, out, host=host, surface_T=288.0, pieces=pieces); ctx.download_wait()
    print("handle + download in %2d pieces: %.3f ms" % (pieces, med(both)))
def dl():
    out.download_async(host, n, 0, 0); ctx.download_wait()
print("download of %d doubles alone: %.3f ms" % (n, med(dl)))
def fold_only():
    col.transmission([False] * 30, out, host=None, surface_T=288.0, pieces=1); ctx.sync()
print("fold alone: %.3f ms" % med(fold_only))
engine.shutdown()
