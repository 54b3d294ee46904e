"""one-off: the sweeps with the library exp vs exp_clamped (scripts/bin/libpyrad_hip_fastexp.so): dump arrays, compare bits"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
out = sys.argv[1]
import bench
from pyrad_amd import _native as nat, engine
cfg, _ = bench.build_workload("C3", 1)
ctx = nat.Context(0)
L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg),
                         cfg["base_resolution"], cfg.get("dynamic_resolution", True))
L.enqueue(surface_T=288.0)
r = L.results()
np.savez(out, **r)
if len(sys.argv) > 2:
    o = np.load(sys.argv[2])
    for k in r:
        d = np.abs(r[k] - o[k]) / np.maximum(np.abs(o[k]), 1e-300)
        print(k, "bit-identical:", np.array_equal(r[k], o[k]), "max rel diff %.3g" % d.max(), "points differing", int((r[k] != o[k]).sum()))
