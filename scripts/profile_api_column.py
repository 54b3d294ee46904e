#!/usr/bin/env python3
"""Where a call of Atmosphere.transmission on the bench column spends its time (host side of the drop-in API): enqueue
(Python + ctypes + the C side's descriptor work) against waiting for the device, beside the engine's column step on the
same box; then a cProfile of one call.  Run on the GPU box."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyrad_amd import model, data, settings, engine, _native as nat
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
marks = {}
orig_wait = nat.Context.download_wait
def wait(self):
    marks["enqueued"] = time.perf_counter()
    orig_wait(self)
    marks["landed"] = time.perf_counter()
nat.Context.download_wait = wait
for _ in range(5):
    ta = time.perf_counter()
    for L in atm: L.changeTemperature(L.T)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); t1 = time.perf_counter()
    print("mutators %.3f ms | call %.3f ms = enqueue %.3f + wait %.3f + return %.3f" % (
        1e3 * (t0 - ta), 1e3 * (t1 - t0), 1e3 * (marks["enqueued"] - t0), 1e3 * (marks["landed"] - marks["enqueued"]), 1e3 * (t1 - marks["landed"])))
nat.Context.download_wait = orig_wait
# the engine's column step (what bench.py times), same box
ctx = model._ctx()
layer_cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
col = engine.ResidentColumn(ctx, layer_cfgs, cfg["surface_T"])
for _ in range(3):
    col.enqueue(layer_arrays=False, merged=True)
ctx.sync()
t0 = time.perf_counter()
for _ in range(10):
    col.enqueue(layer_arrays=False, merged=True)
ctx.sync()
print("engine column step %.3f ms" % (1e2 * (time.perf_counter() - t0)))
t0 = time.perf_counter(); col.enqueue(layer_arrays=False, merged=True); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
print("engine one step: enqueue %.3f ms, then wait %.3f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
col.free()
pr = cProfile.Profile()
for L in atm: L.changeTemperature(L.T)
pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable()
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
print("own us | cumulative us | calls | function")
for own, cum, calls, name in rows[:22]:
    print("%8.0f %8.0f %6d  %s" % (1e6 * own, 1e6 * cum, calls, name))
engine.shutdown()
