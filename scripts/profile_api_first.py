#!/usr/bin/env python3
"""The first Atmosphere.transmission of a freshly built bench column (uploads, buffers, schedules): cProfile by own time.
Run on the GPU box."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyrad_amd import model, data, settings, engine
cfg, _ = bench.build_workload("C5", 1)
c0 = cfg["layers"][0]
settings.set_resolution_multiplier(c0["base_resolution"] / .01)
data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in c0["molecules"]}))
model.Layer.hasAtmosphere = False
engine.get_engine()
for rep in range(2):
    atm = model.Atmosphere("col")
    for c in cfg["layers"]:
        L = atm.addLayer(c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], name=c["name"], dynamicResolution=c.get("dynamic_resolution", True))
        for m in c["molecules"]:
            L.addMolecule(m["species"], **m["conc"])
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable()
    print("first call of column %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t0)))
    st = pstats.Stats(pr)
    rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
    print("own us | cumulative us | calls | function")
    for own, cum, calls, name in rows[:12]:
        print("%8.0f %8.0f %6d  %s" % (1e6 * own, 1e6 * cum, calls, name))
engine.shutdown()
