"""Two independent layer steps in flight: the same resident cell on two contexts (two HIP streams, two
sets of arenas) of one GPU, steps dealt alternately, against all steps on one context.
usage: two_streams.py [C2|C3] [G,r]"""
import sys, os, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from pyrad_amd import _native as nat, engine

workload = sys.argv[1] if len(sys.argv) > 1 else "C3"
shard_of = sys.argv[2] if len(sys.argv) > 2 else None
cfg, desc = bench.build_workload(workload, 1)
mols = bench.molecules_of(cfg)
shard = None
if shard_of:
    G, r = (int(v) for v in shard_of.split(","))
    shard, _ = engine.choose_shards([dict(cfg, molecules=mols)], G, r, "auto")


def make(n_ctx):
    ctxs = [nat.Context(0) for _ in range(n_ctx)]
    layers = [engine.ResidentLayer(c, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                   cfg["base_resolution"], cfg.get("dynamic_resolution", True), shard=shard) for c in ctxs]
    return ctxs, layers


def run(n_ctx, steps=200, reps=3):
    ctxs, layers = make(n_ctx)
    for L in layers:
        L.enqueue(surface_T=288.0)
    for c in ctxs:
        c.sync()
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        for k in range(20):
            layers[k % n_ctx].enqueue(surface_T=288.0)
        for c in ctxs:
            c.sync()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for k in range(steps):
            layers[k % n_ctx].enqueue(surface_T=288.0)
        for c in ctxs:
            c.sync()
        best = min(best, (time.perf_counter() - t0) / steps)
    out = [L.abs_coef.download(L.count if L.plan is not None else L.g["n_work"], L.first if L.plan is not None else 0) for L in layers]
    for L in layers:
        L.free()
    for c in ctxs:
        c.close()
    return best, out


t1, o1 = run(1)
t2, o2 = run(2)
t3, o3 = run(3)
import numpy as np
print("%s %s: 1 stream %.4f ms/step, 2 streams %.4f ms/step (%.1f %%), 3 streams %.4f; results identical: %s"
      % (workload, shard_of or "", t1 * 1e3, t2 * 1e3, (t2 / t1 - 1) * 100, t3 * 1e3,
         all(np.array_equal(o1[0], o) for o in o2 + o3)))
