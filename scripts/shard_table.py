#!/usr/bin/env python3
"""What G-way sharding of a workload costs, measured on ONE GPU: every shard
(G, r) of C3 / C5 is built and stepped alone, exactly what rank r of G computes per step (no
communicator), with equal-width and with cost-balanced bounds.

    python scripts/shard_table.py [C3] [C5] > gpurun_out/shard_table.json

Per (workload, G, mode): t(r) for every r, max_r t(r) (the step of the slowest rank), sum_r t(r),
t_full (G = 1), predicted speed-up t_full / max_r t(r) and the fixed cost per step
(sum_r t(r) - t_full) / (G - 1)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from pyrad_amd import _native as nat, engine  # noqa: E402


def time_steps(ctxs, objs, is_column, steps, merged, budget_s=1.5):
    """objs[i] lives on ctxs[i]: len(objs) independent steps in flight, dealt round-robin"""
    kw = dict(layer_arrays=False, merged=merged) if is_column else dict(surface_T=288.0, merged=merged)
    n = [0]

    def step():
        objs[n[0] % len(objs)].enqueue(**kw)
        n[0] += 1

    def sync():
        for c in ctxs:
            c.sync()
    for _ in objs:
        step()
    sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:                 # clocks
        step()
    sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3


def main():
    workloads = [w for w in sys.argv[1:] if w in ("C2", "C3", "C5")] or ["C3", "C5"]
    in_flight, step_arg = 1, "auto"
    for a in sys.argv[1:]:
        if a.startswith("--in-flight="):
            in_flight = int(a.split("=")[1])
        if a.startswith("--step="):
            step_arg = a.split("=")[1]            # auto (what bench.py --gpus N runs) | merged | per-list
    ctxs = [nat.Context(0) for _ in range(in_flight)]
    ctx = ctxs[0]
    out = {"device": ctx.device_info()["name"], "steps_in_flight_sharded": in_flight, "step": step_arg,
           "source_hash": nat.source_hash(),
           "unit": "ms per step (K1 + K2 + sweep; C5: + column step); t_full with ONE step in flight (what N = 1 runs), "
                   "shards with steps_in_flight_sharded", "rows": []}
    for wl in workloads:
        cfg, desc = bench.build_workload(wl, 1)
        is_column = wl == "C5"
        if is_column:
            layer_cfgs = [dict(c, molecules=bench.molecules_of(c)) for c in cfg["layers"]]
        else:
            layer_cfgs = [dict(cfg, molecules=bench.molecules_of(cfg))]
        steps = 20 if is_column else 100

        def build(shard, ctx=ctx):
            if is_column:
                return engine.ResidentColumn(ctx, layer_cfgs, cfg["surface_T"], shard=shard)
            c = layer_cfgs[0]
            return engine.ResidentLayer(ctx, c["depth"], c["T"], c["P"], c["range_min"], c["range_max"], c["molecules"],
                                        c["base_resolution"], c.get("dynamic_resolution", True), shard=shard)
        full = build(None)
        g0 = (full.layers[0] if is_column else full).g
        n_lists, n_layers = len(full.jobs), (len(full.layers) if is_column else 1)
        step_of = lambda G: bench.choose_step(step_arg, g0["n_work"], n_lists, n_layers, G)
        t_full = time_steps([ctx], [full], is_column, steps, step_of(1))
        evals_full = full.evals
        full.free()
        print("%s full: %.4f ms" % (wl, t_full), file=sys.stderr)
        for G in (2, 4, 8):
            for mode in ("equal", "balanced"):
                ts, ev, bounds = [], 0, None
                for r in range(G):
                    plan = engine.balanced_shards(layer_cfgs, G, r) if mode == "balanced" else engine.as_plan((G, r), full.n)
                    bounds = plan.bounds
                    parts = [build(plan, c) for c in ctxs]
                    ts.append(time_steps(ctxs, parts, is_column, steps, step_of(G)))
                    ev += parts[0].evals
                    for part in parts:
                        part.free()
                row = dict(workload=wl, G=G, bounds=mode, step="merged" if step_of(G) else "per-list", t_full_ms=t_full, t_r_ms=ts, max_ms=max(ts), sum_ms=sum(ts),
                           predicted_speedup=t_full / max(ts), fixed_ms_per_step=(sum(ts) - t_full) / (G - 1),
                           imbalance=max(ts) / (sum(ts) / G), evals_match=bool(ev == evals_full),
                           shard_points=[c for _, c in bounds])
                out["rows"].append(row)
                print("%s G=%d %-8s max %.4f sum %.4f speed-up %.2f imbalance %.3f fixed %.4f" % (
                    wl, G, mode, row["max_ms"], row["sum_ms"], row["predicted_speedup"], row["imbalance"],
                    row["fixed_ms_per_step"]), file=sys.stderr)
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
