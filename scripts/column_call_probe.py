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
# the call's phases: entry -> the C entry point returns (everything enqueued) -> bookkeeping done -> the spectrum has landed
from pyrad_amd import _native as nat
marks = {}
orig_t, orig_w = nat.Column.transmission, nat.Context.download_wait
def t_wrap(self, *a, **k):
    marks["t_in"] = time.perf_counter(); orig_t(self, *a, **k); marks["t_out"] = time.perf_counter()
def w_wrap(self):
    marks["w_in"] = time.perf_counter(); orig_w(self); marks["w_out"] = time.perf_counter()
nat.Column.transmission, nat.Context.download_wait = t_wrap, w_wrap
rows = []
for _ in range(9):
    for L in atm: L.changeTemperature(L.T)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); t1 = time.perf_counter()
    rows.append((marks["t_in"] - t0, marks["t_out"] - marks["t_in"], marks["w_in"] - marks["t_out"], marks["w_out"] - marks["w_in"], t1 - marks["w_out"], t1 - t0))
r = np.median(np.array(rows), axis=0) * 1e3
print("phases (ms): before the C call %.3f | C call %.3f | bookkeeping %.3f | wait %.3f | return %.3f | total %.3f" % tuple(r))
nat.Column.transmission, nat.Context.download_wait = orig_t, orig_w
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
import cProfile, pstats
pr = cProfile.Profile()
for _ in range(20):
    for L in atm: L.changeTemperature(L.T)
    pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable()
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
print("per call: own us | cumulative us | calls | function")
for own, cum, calls, name in rows[:16]:
    print("%8.1f %8.1f %6.1f  %s" % (1e6 * own / 20, 1e6 * cum / 20, calls / 20, name))
engine.shutdown()
