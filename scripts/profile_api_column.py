#!/usr/bin/env python3
"""cProfile of Atmosphere.transmission on the bench column (host side of the drop-in API); run on the GPU box."""
import cProfile, pstats, io, os, sys, time
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
for _ in range(3):
    for L in atm: L.changeTemperature(L.T)
    t0 = time.perf_counter(); atm.transmission(surfaceTemperature=288); print("call %.3f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
for L in atm: L.changeTemperature(L.T)
pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue())
engine.shutdown()
