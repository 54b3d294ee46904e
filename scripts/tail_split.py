#!/usr/bin/env python3
"""Does the tail of the accumulate launch pay for a finer split?  Merged C3 has 9,375 spans = 2.29 rounds of the chip's
4,096 wave slots: time the first `head` spans unsplit and the rest with the lines of a span shared by 1 / 2 / 4 waves,
as two separate steps over two grid ranges (their sum bounds what one mixed launch could reach from above: the two
do not overlap here).  A development aid (GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyrad_amd import _native as nat, engine                     # noqa: E402
from pyrad_amd.dist import ShardPlan                              # noqa: E402
import bench                                                      # noqa: E402


def time_steps(ctx, fn, n=40):
    for _ in range(5):
        fn()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e3


def main():
    heads = [int(a) for a in sys.argv[1:]] or [8192]
    cfg, desc = bench.build_workload("C3", 1)
    mols = bench.molecules_of(cfg)
    for head in heads:
        for split in (1, 2, 4):
            res = []
            for rank in (0, 1):
                ctx = nat.Context(0)
                ctx.set_option("accum_line_split", 1 if rank == 0 else split)
                L0 = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                          cfg["base_resolution"], cfg.get("dynamic_resolution", True))
                n = L0.n
                L0.free()
                plan = ShardPlan(n, [(0, head * 256), (head * 256, n - head * 256)], rank)
                L = engine.ResidentLayer(ctx, cfg["depth"], cfg["T"], cfg["P"], cfg["range_min"], cfg["range_max"], mols,
                                         cfg["base_resolution"], cfg.get("dynamic_resolution", True), shard=plan)
                t = time_steps(ctx, lambda: L.enqueue(surface_T=288.0, merged=True))
                res.append(t)
                L.free()
                ctx.close()
            print("head %d spans unsplit %.4f ms + tail split %d %.4f ms = %.4f ms" % (head, res[0], split, res[1], sum(res)), flush=True)


if __name__ == "__main__":
    main()
