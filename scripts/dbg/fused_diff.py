import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pyrad_amd import _native as nat, engine, synthetic
import bench
ctx = nat.Context(0)
cfg = synthetic.config_c3()
mols = bench.molecules_of(cfg)
args = (cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols, cfg["base_resolution"], cfg["dynamic_resolution"])
for shard in (None, engine.balanced_shards([dict(cfg, molecules=mols)], 8, 0), (8, 0)):
    L = engine.ResidentLayer(ctx, *args, shard=shard)
    sl = slice(L.first, L.first + L.count)
    L.enqueue(surface_T=288, fused=True)
    a = {k: v[sl].copy() for k, v in L.results().items()}
    xa = [L.xsec_host(i)[sl].copy() for i in range(3)]
    for b in [L.abs_coef, L.trans, L.I_out] + [j[3] for j in L.jobs]:
        b.fill(0.0)
    L.enqueue(surface_T=288, fused=False)
    b_ = {k: v[sl].copy() for k, v in L.results().items()}
    xb = [L.xsec_host(i)[sl].copy() for i in range(3)]
    print("shard", None if shard is None else (L.first, L.count))
    for i in range(3):
        d = np.nonzero(xa[i] != xb[i])[0]
        print(" xsec", i, "mismatch", d.size, d[:8])
    for k in a:
        d = np.nonzero(a[k] != b_[k])[0]
        rel = np.abs(a[k] - b_[k]) / np.maximum(np.abs(b_[k]), 1e-300)
        print(" ", k, "mismatch", d.size, "max rel", rel.max(), "first idx", d[:10], "mod1024", (d[:10] % 1024))
    L.free()
