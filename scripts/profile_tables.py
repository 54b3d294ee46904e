#!/usr/bin/env python3
"""Tables of DESIGN.md §4 from the committed profile set: `python scripts/profile_tables.py r05`.
Reads profiles/<tag>_<workload>_{bench,pmc}.json and <tag>_<workload>_kernel_stats.csv; prints markdown."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"


def load(wl, kind):
    f = os.path.join(P, "%s_%s_%s.json" % (tag, wl, kind))
    return json.load(open(f)) if os.path.isfile(f) else None


def kstats(wl):
    f = os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, wl))
    out = {}
    if os.path.isfile(f):
        for line in open(f).read().splitlines()[1:]:
            name, calls, total, avg, mn, mx = line.rsplit(",", 5)      # (template arguments carry commas of their own)
            out[name] = (int(calls), float(avg))
    return out


def find(ks, sub):
    return [(k, v) for k, v in ks.items() if sub in k]


print("## steps\n")
print("| workload | step (blocks) | K1 | K2 (rocprofv3 avg) | sweep | evals/s | K2 VALU busy | budget | other step | traffic / algorithmic |")
print("|---|---|---|---|---|---|---|---|---|---|")
for wl in ("C1", "C2", "C3", "C5", "C3s8", "C3s4", "C3s2"):
    d = load(wl, "bench")
    if d is None:
        continue
    ks = kstats(wl)
    k1 = sum(v[1] * v[0] for k, v in ks.items() if "line_prep" in k) / max(1, max([v[0] for k, v in ks.items() if "line_prep" in k] or [1]))
    k2 = {k: v for k, v in ks.items() if "xsec_accumulate" in k}
    sw = {k: v for k, v in ks.items() if "sweep" in k or "column_step" in k}
    other = d.get("per_list_leg") or d.get("merged_leg") or {}
    r = d["roofline"]
    print("| %s | %.4f (%s) | %.1f us | %s | %s | %.3g | %s | %s | %s | %s |" % (
        wl, d["ms_per_step"], ", ".join("%.4f" % b for b in d["ms_per_step_blocks"]["blocks"][1:]), k1,
        " + ".join("%s %.1f us" % (k.replace("xsec_accumulate_", ""), v[1]) for k, v in k2.items()),
        " + ".join("%s %.1f us" % (k.split("<")[0], v[1]) for k, v in sw.items()) or "fused",
        d["value"], ("%.3f" % d["valu_f64"]["busy_frac"]) if d["valu_f64"].get("busy_frac") else "-",
        ("%.4f" % d["budget_leg"]["ms_per_step"]) if "budget_leg" in d else "-",
        ("%s %.4f" % (other.get("step"), other["ms_per_step"])) if other else "-",
        ("%.1f / %.1f MB = %.2f" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6, r["traffic"] / r["algorithmic_bytes_per_launch"])) if r.get("traffic") else "-"))
    print("  config.step:", d["config"]["step"][:60], "| roofline.frac %.4f" % r["frac"], "| stale", r.get("traffic_stale"))
    if "api_path" in d:
        a = d["api_path"]
        print("  api:", {k: round(v, 3) for k, v in a.items() if isinstance(v, float) and k.startswith("ms_")})
    if "in_flight_leg" in d:
        print("  in flight:", d["in_flight_leg"]["steps_in_flight"], "%.4f ms" % d["in_flight_leg"]["ms_per_step"])
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("  cpu: python %.3g, C %.3g, numpy %.3g evals/s; whole workload C port %.1f s, python %.0f s" % (
            c["value"], c["c_port_value"], c["vectorised_value"], c["at_survey_extent"]["whole_workload_seconds_c_port"],
            c["at_survey_extent"]["whole_workload_seconds_python"]))

print("\n## where the accumulate kernel's wave-cycles go\n")
print("| shape | kernel | waves | WAVE_CYCLES | ACTIVE_INST_ANY | of it VALU | WAIT_INST_ANY (issue stall) | WAIT_ANY (waitcnt / barrier) | VALU / SALU / LDS / SMEM / VMEM instructions | LDS bank-conflict cycles |")
print("|---|---|---|---|---|---|---|---|---|---|")
for wl in ("C3", "C3s8", "C5"):
    p = load(wl, "pmc")
    if p is None:
        continue
    for k, v in p["kernels"].items():
        if "xsec_accumulate" not in k or "valu" not in v:
            continue
        c = v["valu"]
        wc = c.get("SQ_WAVE_CYCLES")
        if not wc:
            continue
        pct = lambda x: "%.3g (%.1f %%)" % (c.get(x, 0.0), 100.0 * c.get(x, 0.0) / wc)
        print("| %s | %s | %d | %.4g | %s | %s | %s | %s | %.4g / %.3g / %.3g / %.3g / %.3g | %.3g |" % (
            wl, k, round(c.get("SQ_WAVES", 0)), wc, pct("SQ_ACTIVE_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"), pct("SQ_WAIT_INST_ANY"),
            pct("SQ_WAIT_ANY"), c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_SMEM", 0),
            c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0), c.get("SQ_LDS_BANK_CONFLICT", 0)))
