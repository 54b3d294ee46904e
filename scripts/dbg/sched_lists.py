"""Dump the device-built dispatch lists (and span tables) of a few launches, to compare two library builds:
   PYRAD_HIP_LIB=... python3 scripts/dbg/sched_lists.py <out dir>"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pyrad_amd import _native as nat, engine, synthetic

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
ctx = nat.Context(0)
# the 30-layer column: two groups (far-field kernel, skewed-range kernel), merged jobs
cfg = synthetic.config_c5(n_layers=30, n_lines=131072)
col = engine.ResidentColumn(ctx, [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]], cfg["surface_T"])
col.enqueue(merged=True)
ctx.sync()
for k in range(2):
    lst, tabs, on_dev = ctx.schedule_export(k)
    assert on_dev
    np.save(os.path.join(out, "C5_list%d.npy" % k), lst); np.save(os.path.join(out, "C5_tabs%d.npy" % k), tabs)
t0 = time.perf_counter()
# a lopsided cell: one part of the XCD partition holds most of the tiles (beyond what the rank kernel takes)
base = synthetic.make_lines(77, 60000, 1000.0, 1002.0)
mol = dict(conc=4e-4, isotopologues=[dict(lines=base, molmass=synthetic.SPECIES["co2"]["molmass"],
                                          q_T=synthetic.q_value("co2", 296), q296=synthetic.SPECIES["co2"]["q296"])])
L = engine.ResidentLayer(ctx, 10.0, 296, 60.0, 100.0, 2500.0, [mol] * 8, .0001, False)
L.enqueue(surface_T=288.0)
ctx.sync()
lst, tabs, on_dev = ctx.schedule_export(0)
assert on_dev
np.save(os.path.join(out, "lopsided_list.npy"), lst)
# the merged 100-2500 cm^-1 cell
cfg3 = synthetic.config_c3(n_lines=131072)
L3 = engine.ResidentLayer(ctx, cfg3["depth"], cfg3["T"], cfg3["P"], cfg3["range_min"], cfg3["range_max"], bench.molecules_of(cfg3),
                          cfg3["base_resolution"], False)
L3.enqueue(surface_T=288.0, merged=True)
ctx.sync()
lst, tabs, on_dev = ctx.schedule_export(0)
np.save(os.path.join(out, "C3_list.npy"), lst)
print("dumped to", out)
