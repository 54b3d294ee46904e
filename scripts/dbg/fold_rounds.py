"""Is the column fold's time quantised in rounds of resident waves?  30 layers of n points, the fold over counts around the
multiples of one round (4 points per thread, 256 threads per block, 256 CUs x 16 resident waves at four per SIMD)."""
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
one_round = 256 * 16 * 64 * 4            # points of one round at four waves per SIMD
for count in [one_round, 2 * one_round, n]:
    for _ in range(20):
        ctx.column_fold_dev(ks, T, depth, 100.0, 2500.0, n, out, surface_T=288.0, first=0, count=count)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.column_fold_dev(ks, T, depth, 100.0, 2500.0, n, out, surface_T=288.0, first=0, count=count)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 50
    print("count %8d (%.2f rounds): %.1f us, %.2f TB/s" % (count, count / one_round, dt * 1e6, count * 8 * 31 / dt / 1e12))
