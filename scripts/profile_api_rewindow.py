#!/usr/bin/env python3
"""Where Atmosphere.transmission spends its time right after changePressure on every layer of the bench column (new
windows: new line selections, new merged orders and dispatch schedules, built on the device).  Run on the GPU box."""
import cProfile, pstats, os, sys, time
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
for rep in range(6):
    f = 0.99 if rep % 2 == 0 else 1.0
    ta = time.perf_counter()
    for L, c in zip(atm, cfg["layers"]):
        L.changePressure(c["P"] * f)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); t1 = time.perf_counter()
    print("mutators %.3f ms | call %.3f ms = enqueue %.3f + wait %.3f" % (
        1e3 * (t0 - ta), 1e3 * (t1 - t0), 1e3 * (marks["enqueued"] - t0), 1e3 * (marks["landed"] - marks["enqueued"])))
nat.Context.download_wait = orig_wait
def table(pr, n=18):
    st = pstats.Stats(pr)
    rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
    print("own us | cumulative us | calls | function")
    for own, cum, calls, name in rows[:n]:
        print("%8.0f %8.0f %6d  %s" % (1e6 * own, 1e6 * cum, calls, name))
pm = cProfile.Profile()
pm.enable()
for L, c in zip(atm, cfg["layers"]):
    L.changePressure(c["P"] * 0.98)
pm.disable()
print("--- the 30 changePressure calls"); table(pm, 14)
pr = cProfile.Profile()
pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable()
print("--- the call after them")
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
print("own us | cumulative us | calls | function")
for own, cum, calls, name in rows[:25]:
    print("%8.0f %8.0f %6d  %s" % (1e6 * own, 1e6 * cum, calls, name))
engine.shutdown()
