"""Diagnostic (GPU): host enqueue time per C2 step against the end-to-end time per step, with and
without the one-rank RCCL all-gather pipeline (no profiling events): is the loop GPU- or host-bound?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["x"]
import bench
from pyrad_amd import _native as nat, engine
ctx = nat.Context(0)
cfg, _ = bench.build_workload("C2", 1)
Ls = [engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], bench.molecules_of(cfg), cfg["base_resolution"], False) for _ in range(2)]
comm = nat.Comm(ctx, nat.Comm.unique_id(), 1, 0)
for L in Ls: L.enqueue(surface_T=288.0)
ctx.sync()
def run(n, use_comm):
    t0 = time.perf_counter()
    for k in range(n):
        L = Ls[k % 2]
        if use_comm: comm.fence_dev(k % 2)
        L.enqueue(surface_T=288.0)
        if use_comm: L.enqueue_allgather(comm, (L.abs_coef,), overlap_slot=k % 2)
    t1 = time.perf_counter()
    if use_comm: comm.fence_dev(-1)
    ctx.sync()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for use in (False, True):
    run(50, use)
    h, tot = run(400, use)
    print("comm" if use else "nocomm", "host enqueue us/step %.1f" % h, "total us/step %.1f" % tot)
