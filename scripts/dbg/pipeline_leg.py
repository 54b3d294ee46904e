"""one-off: C3 (or another cell) with 1 step in flight vs 2 unordered vs 2 / 3 chained (lbl_ctx_chain_accumulate)"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, _ = bench.build_workload(w, 1)
for n, ch in ((1, False), (2, False), (2, True), (3, True), (1, False)):
    r = bench.in_flight_leg(cfg, n_flight=n, steps=100, chained=ch)
    print(w, n, ch, round(r["ms_per_step"], 5), "%.3e" % r["evals_per_s"], flush=True)
