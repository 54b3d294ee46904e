"""Where the time of one model.getAbsCoef call goes (C2 / C3): host work, kernels, download."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from pyrad_amd import model, data, settings, engine, synthetic
for wl in ("C2", "C3"):
    cfg, _ = bench.build_workload(wl, 1)
    settings.set_resolution_multiplier(cfg["base_resolution"] / .01)
    data.set_source(data.synthetic_source({m["species"]: m["lines"] for m in cfg["molecules"]}))
    model.Layer.hasAtmosphere = False
    layer = model.Layer(cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], dynamicResolution=False)
    for m in cfg["molecules"]:
        layer.addMolecule(m["species"], **m["conc"])
    k = model.getAbsCoef(layer)
    ctx = engine.get_engine().ctx
    n = k.size
    T = {}
    for rep in range(7):
        t0 = time.perf_counter(); layer.changeTemperature(cfg["T"]); t1 = time.perf_counter()
        st, _ = layer._ensure_swept(); t2 = time.perf_counter()
        ctx.sync(); t3 = time.perf_counter()
        a = st.bufs["abs_coef"].download(n, pinned=True); t4 = time.perf_counter()
        b = st.bufs["abs_coef"].download(n); t5 = time.perf_counter()
        surf = layer.planck(288); t6 = time.perf_counter()
        spec = layer.transmission(surf); t7 = time.perf_counter()
        spec2 = layer.transmission(np.array(surf)); t8 = time.perf_counter()
        for name, dt in (("reset", t1 - t0), ("enqueue(host)", t2 - t1), ("kernels(sync)", t3 - t2), ("download pinned", t4 - t3),
                         ("download pageable", t5 - t4), ("planck", t6 - t5), ("transmission(pinned in)", t7 - t6),
                         ("transmission(pageable in)", t8 - t7)):
            T.setdefault(name, []).append(dt * 1e3)
    print(wl, n, {k_: round(float(np.median(v)), 4) for k_, v in T.items()})
    engine.shutdown()
