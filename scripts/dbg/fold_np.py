"""The column fold with four points per thread (first point a multiple of 4) against two (first = 2): 30 layers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyrad_amd import _native as nat
ctx = nat.Context(0)
n = 2400000
rng = np.random.default_rng(1)
ks = [ctx.buffer(n).upload(rng.uniform(1e-6, 1e-3, n)) for _ in range(30)]
T = list(np.linspace(288, 217, 30).round())
depth = [1e4] * 30
out = ctx.buffer(n)
for first in (0, 2, 0, 2):
    count = n - 4
    for _ in range(20):
        ctx.column_fold_dev(ks, T, depth, 100.0, 2500.0, n, out, surface_T=288.0, first=first, count=count)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.column_fold_dev(ks, T, depth, 100.0, 2500.0, n, out, surface_T=288.0, first=first, count=count)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 50
    print("first %d: %.1f us" % (first, dt * 1e6))
