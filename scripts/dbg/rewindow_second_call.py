import cProfile, pstats, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
for rep in range(4):
    f = 0.99 if rep % 2 == 0 else 1.0
    for L, c in zip(atm, cfg["layers"]):
        L.changePressure(c["P"] * f)
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); atm.transmission(surfaceTemperature=288); pr.disable(); t1 = time.perf_counter()
    print("rep", rep, "call %.3f ms" % (1e3 * (t1 - t0)))
    st = pstats.Stats(pr)
    rows = sorted(((v[2], v[3], v[0], "%s:%d %s" % (os.path.basename(k[0]), k[1], k[2])) for k, v in st.stats.items()), reverse=True)
    for own, cum, calls, name in rows[:5]:
        print("   %8.0f %8.0f %6d  %s" % (1e6 * own, 1e6 * cum, calls, name))
engine.shutdown()
