"""Atmosphere.transmission after changePressure on EVERY layer, each time to pressures not seen before (new windows, new
line selections, new schedules), at sustained clocks; prints the call times and a cProfile of the last calls.  Run with
LBL_TRACE=1 for the library's own phase lines."""
import cProfile, pstats, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for _ in range(60):
    for L in atm: L.changeTemperature(L.T)
    atm.transmission(surfaceTemperature=288)
t_res = []
for _ in range(5):
    for L in atm: L.changeTemperature(L.T)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); t_res.append(time.perf_counter() - t0)
print("resident call: %.3f ms" % (1e3 * float(np.median(t_res))))
pr = cProfile.Profile()
t_mut, t_call = [], []
N = 8
for rep in range(N):
    f = 0.99 - 0.003 * rep
    t0 = time.perf_counter()
    for L, c in zip(atm, cfg["layers"]):
        L.changePressure(c["P"] * f)
    t_mut.append(time.perf_counter() - t0)
    sys.stderr.write("rep %d\n" % rep)
    if rep >= N - 4: pr.enable()
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); t1 = time.perf_counter()
    if rep >= N - 4: pr.disable()
    t_call.append(t1 - t0)
print("after changePressure on all layers: calls", [round(1e3 * t, 3) for t in t_call], "ms; mutators", [round(1e3 * t, 3) for t in t_mut])
st = pstats.Stats(pr)
rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
print("per call: own us | cumulative us | calls | function")
for own, cum, calls, name in rows[:22]:
    print("%8.1f %8.1f %6.1f  %s" % (1e6 * own / 4, 1e6 * cum / 4, calls / 4, name))
engine.shutdown()
