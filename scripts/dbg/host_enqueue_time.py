"""Host cost of enqueueing one step (Python + ctypes + the C++ descriptor work), measured on a grid so
small that the GPU keeps up: three line lists, sharded plan, with and without the communicator calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrad_amd import _native as nat, engine, synthetic
ctx = nat.Context(0)
g = engine.layer_grid(1013.25, 600, 601, .001, False)
mols = []
for s_, seed, conc in (("co2", 61, 4e-4), ("h2o", 62, 1e-2), ("ch4", 63, 1.8e-6)):
    sp = synthetic.SPECIES[s_]
    mols.append(dict(conc=conc, isotopologues=[dict(lines=synthetic.make_lines(seed, 50, g["eff_min"], g["eff_max"]),
                                                    molmass=sp["molmass"], q_T=synthetic.q_value(s_, 296), q296=sp["q296"])]))
for n_mol in (1, 3):
    L = engine.ResidentLayer(ctx, 10.0, 296, 1013.25, 600, 601, mols[:n_mol], .001, False)
    L.enqueue(surface_T=288.0); ctx.sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(2000):
            L.enqueue(surface_T=288.0)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
    print("%d line list(s): host %.1f us per enqueue (+ %.1f us drain per step)" % (n_mol, (t1 - t0) / 2000 * 1e6, (t2 - t1) / 2000 * 1e6))
    L.free()
