"""Diagnostic (GPU): time to first spectrum through the drop-in API, and after re-windowing
(changePressure / changeRange re-read the lines and move the window: pyradClasses.py:734-752 -> resetData,
cls:45-56).  With LBL_TRACE=1 the library prints its host-side setup phases.  cProfile of the first call on request.

    python scripts/first_call.py C3 [--profile]
    python scripts/first_call.py C5
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from pyrad_amd import model, data, settings, engine


def t(fn):
    t0 = time.perf_counter()
    r = fn()
    return (time.perf_counter() - t0) * 1e3, r


def cell(wl, profile):
    cfg, _ = bench.build_workload(wl, 1)
    settings.set_resolution_multiplier(cfg["base_resolution"] / .01)
    data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in cfg["molecules"]}))
    model.Layer.hasAtmosphere = False
    engine.get_engine()                      # context creation is not part of any leg
    if "--warm" in sys.argv:                 # the HIP runtime's own first-use costs (first hipMalloc, first copies) paid by a small cell
        w = model.Layer(10.0, 296, 1013.25, 600, 601, dynamicResolution=False)
        w.addMolecule(cfg["molecules"][0]["species"], ppm=400)
        model.getAbsCoef(w)
        print("---- warmed up", file=sys.stderr)

    def build():
        layer = model.Layer(cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], dynamicResolution=False)
        for m in cfg["molecules"]:
            layer.addMolecule(m["species"], **m["conc"])
        return layer
    ms_build, layer = t(build)
    if profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    print("---- first getAbsCoef", file=sys.stderr)
    ms_first, k = t(lambda: model.getAbsCoef(layer))
    if profile:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    out = {"build": ms_build, "first": ms_first}
    for rep in range(3):
        layer.changeTemperature(cfg["T"])
        out.setdefault("after_changeTemperature", []).append(t(lambda: model.getAbsCoef(layer))[0])
    for rep, P in enumerate((900.0, 800.0, 1013.25)):
        print("---- changePressure %s" % P, file=sys.stderr)
        ms_m, _ = t(lambda: layer.changePressure(P))
        ms_g, _ = t(lambda: model.getAbsCoef(layer))
        out.setdefault("changePressure", []).append((round(ms_m, 3), round(ms_g, 3)))
    for rep, (a, b) in enumerate(((200, 2400), (300, 2300), (cfg["range_min"], cfg["range_max"]))):
        print("---- changeRange %s %s" % (a, b), file=sys.stderr)
        ms_m, _ = t(lambda: layer.changeRange(a, b))
        ms_g, _ = t(lambda: model.getAbsCoef(layer))
        out.setdefault("changeRange", []).append((round(ms_m, 3), round(ms_g, 3)))
    print(wl, {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in out.items()})
    engine.shutdown()


def column(profile):
    cfg, _ = bench.build_workload("C5", 1)
    c0 = cfg["layers"][0]
    settings.set_resolution_multiplier(c0["base_resolution"] / .01)
    data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in c0["molecules"]}))
    model.Layer.hasAtmosphere = False
    engine.get_engine()

    def build():
        atm = model.Atmosphere("column")
        for c in cfg["layers"]:
            L = atm.addLayer(c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], name=c["name"], dynamicResolution=False)
            for m in c["molecules"]:
                L.addMolecule(m["species"], **m["conc"])
        return atm
    if profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
    ms_build, atm = t(build)
    if profile:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
        pr = cProfile.Profile(); pr.enable()
    print("---- first transmission", file=sys.stderr)
    ms_first, spec = t(lambda: atm.transmission(surfaceTemperature=cfg["surface_T"]))
    if profile:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    out = {"build": ms_build, "first": ms_first}
    for rep in range(3):
        for L in atm:
            L.changeTemperature(L.T)
        out.setdefault("after_changeTemperature", []).append(round(t(lambda: atm.transmission(surfaceTemperature=cfg["surface_T"]))[0], 3))
    for rep in range(2):
        print("---- changePressure of every layer", file=sys.stderr)
        ms_m, _ = t(lambda: [L.changePressure(L.P * (0.99 if rep == 0 else 1 / 0.99)) for L in atm])
        ms_g, _ = t(lambda: atm.transmission(surfaceTemperature=cfg["surface_T"]))
        out.setdefault("changePressure", []).append((round(ms_m, 3), round(ms_g, 3)))
    print("C5", {k_: (round(v, 3) if isinstance(v, float) else v) for k_, v in out.items()})
    engine.shutdown()


if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
    prof = "--profile" in sys.argv
    if wl == "C5":
        column(prof)
    else:
        cell(wl, prof)
