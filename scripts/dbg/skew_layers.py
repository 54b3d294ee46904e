"""Timing of a column made of a SUBSET of config 5's layers (which of them run the skewed-range kernel, with which line split):
   python3 scripts/dbg/skew_layers.py <first layer> <last layer, exclusive> [KEY=VALUE options ...]
prints the step time (wall clock over 40 merged steps, one sync) and the event-timed kernel classes."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench                                   # noqa: E402
from pyrad_amd import _native as nat, engine, synthetic   # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
cfg = synthetic.config_c5(n_layers=30, n_lines=131072)
cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"][lo:hi]]
ctx = nat.Context(0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
col = engine.ResidentColumn(ctx, cfgs, cfg["surface_T"])
for _ in range(30):
    col.enqueue(merged=True)
ctx.sync()
t0 = time.perf_counter()
for _ in range(40):
    col.enqueue(merged=True)
ctx.sync()
wall = (time.perf_counter() - t0) / 40
ctx.profile_enable(True)
ctx.profile_reset()
for _ in range(10):
    col.enqueue(merged=True)
ctx.sync()
prof = ctx.profile_read()
print("layers %d..%d %s: step %.4f ms" % (lo, hi, " ".join(sys.argv[3:]), wall * 1e3),
      {k: round(v[1] / 10, 4) for k, v in prof.items() if v[0]})
