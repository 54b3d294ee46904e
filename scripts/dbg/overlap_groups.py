"""What would running the column's two accumulate launches (far-field kernel: the 13 wide layers; skewed-range kernel: the 17
narrow ones) side by side on two streams gain?  The two halves of config 5 as two columns on two contexts of one GPU: each
alone, one after the other, and both in flight."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from pyrad_amd import _native as nat, engine, synthetic

cfg = synthetic.config_c5(n_layers=30, n_lines=131072)
cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
split = int(sys.argv[1]) if len(sys.argv) > 1 else 13
ca, cb = nat.Context(0), nat.Context(0)
A = engine.ResidentColumn(ca, cfgs[:split], cfg["surface_T"])
B = engine.ResidentColumn(cb, cfgs[split:], cfg["surface_T"])


def timed(fn, sync, reps=40):
    for _ in range(30):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps * 1e3


both_sync = lambda: (ca.sync(), cb.sync())
ta = timed(lambda: A.enqueue(merged=True), ca.sync)
tb = timed(lambda: B.enqueue(merged=True), cb.sync)


def serial():
    A.enqueue(merged=True); ca.sync(); B.enqueue(merged=True); cb.sync()


def concurrent():
    A.enqueue(merged=True); B.enqueue(merged=True)


ts = timed(serial, both_sync)
tc = timed(concurrent, both_sync)
print("wide layers alone %.4f ms, narrow layers alone %.4f ms, sum %.4f; one after the other (host syncs) %.4f; both in flight %.4f ms"
      % (ta, tb, ta + tb, ts, tc))
