#!/usr/bin/env python3
"""Reduce the rocprofv3 output of profiles/collect.sh to small committed summaries:
  <tag>_<workload>_kernel_stats.csv   per-kernel calls / total / average duration (kernel trace)
  <tag>_<workload>_pmc.json           per-kernel HBM bytes per launch from FETCH_SIZE / WRITE_SIZE
  pmc_traffic.json                    what bench.py reports as roofline.traffic (C2 only)
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide (16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM section): the corrected figure
doubles it; narrower access widths are uncalibrated, so both raw and corrected values are kept."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "").replace("lbl::", "")
    return name.split("(")[0]


def kernel_trace(d):
    f = find(d, "kernel_trace.csv")
    agg = collections.OrderedDict()
    if not f:
        return agg
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3      # us
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    return agg


def counters(d):
    f = find(d, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    if not f:
        return agg, {}
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}


def main():
    out, tag, wl = sys.argv[1], sys.argv[2], sys.argv[3]
    here = os.path.dirname(os.path.abspath(__file__))
    kt = kernel_trace(os.path.join(out, "trace_" + wl))
    with open(os.path.join(here, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as f:
        f.write("kernel,calls,total_us,avg_us,min_us,max_us\n")
        for k, (c, tot, mn, mx) in sorted(kt.items(), key=lambda kv: -kv[1][1]):
            f.write("%s,%d,%.3f,%.3f,%.3f,%.3f\n" % (k, c, tot, tot / c, mn, mx))
    raw = find(os.path.join(out, "trace_" + wl), "kernel_stats.csv")          # rocprofv3 --stats output, verbatim
    if raw:
        with open(raw) as f, open(os.path.join(here, "%s_%s_rocprofv3_stats.csv" % (tag, wl)), "w") as g:
            g.write(f.read())
    fetch, nf = counters(os.path.join(out, "pmc_fetch_" + wl))
    write, nw = counters(os.path.join(out, "pmc_write_" + wl))
    valu, nv = counters(os.path.join(out, "pmc_valu_" + wl))
    for extra in ("pmc_sq2_", "pmc_sq3_"):           # round 5: the wave-cycle buckets and the instruction mix (same command, own passes)
        more, nm = counters(os.path.join(out, extra + wl))
        for k, c in more.items():
            for name, total in c.items():
                # (per-launch means like the first pass's: scaled to that pass's launch count)
                valu[k][name] = total / max(nm.get(k, 1), 1) * max(nv.get(k, nm.get(k, 1)), 1)
            nv.setdefault(k, nm.get(k, 1))
    pmc = {"units": "bytes per launch", "workload": wl, "note": __doc__.split("FETCH_SIZE / WRITE_SIZE")[1].strip(),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fb = fetch[k].get("FETCH_SIZE", 0.0) * 1024 / max(nf.get(k, 1), 1)
        wb = write[k].get("WRITE_SIZE", 0.0) * 1024 / max(nw.get(k, 1), 1)
        pmc["kernels"][k] = {"fetch_raw": fb, "fetch_x2": 2 * fb, "write": wb, "hbm_bytes_corrected": 2 * fb + wb,
                             "launches_fetch_pass": nf.get(k, 0), "launches_write_pass": nw.get(k, 0)}
        if k in valu:
            c = valu[k]; nl = max(nv.get(k, 1), 1)
            pmc["kernels"][k]["valu"] = {kk: vv / nl for kk, vv in c.items()}
    with open(os.path.join(here, "%s_%s_pmc.json" % (tag, wl)), "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    # what bench.py reports as roofline.traffic, keyed by workload
    tpath = os.path.join(here, "pmc_traffic.json")
    allt = {}
    if os.path.isfile(tpath):
        try:
            allt = json.load(open(tpath))
        except ValueError:
            allt = {}
    if "hbm_bytes_per_launch" in allt:          # old single-workload layout
        allt = {}
    traffic, valu_insts, avg_us, weights = {}, {}, {}, {}
    for k, v in pmc["kernels"].items():
        # collect.sh profiles bench.py --no-direct-pass, so every xsec_accumulate_* launch belongs to the
        # timed path: the far-field kernel <R, LS, true> and, for a column, the all-direct instantiations
        # <R, LS, false> its narrow-window layer groups take; launch-weighted means over all of them
        if k.startswith("xsec_accumulate"):
            base = "xsec_accumulate_kernel"
        else:
            base = k.split("<")[0]
        # several instantiations can share a base name (a column launches one kernel per window
        # group): launch-weighted means
        w = max(v.get("launches_fetch_pass", 1), 1)
        acc = weights.setdefault(base, [0.0, 0.0, 0.0, 0.0, 0.0])
        acc[0] += w; acc[1] += w * v["hbm_bytes_corrected"]
        if "valu" in v and "SQ_INSTS_VALU" in v["valu"]:
            acc[2] += w * v["valu"]["SQ_INSTS_VALU"]
        if k in kt:
            acc[3] += kt[k][0]; acc[4] += kt[k][1]
    for base, acc in weights.items():
        traffic[base] = acc[1] / acc[0]
        if acc[2]:
            valu_insts[base] = acc[2] / acc[0]
        if acc[3]:
            avg_us[base] = acc[4] / acc[3]
    sys.path.insert(0, os.path.dirname(here))
    from pyrad_amd import _native
    allt[wl] = {"source": "%s_%s_pmc.json" % (tag, wl), "source_hash": _native.source_hash(),
                "hbm_bytes_per_launch": traffic, "valu_wave_insts_per_launch": valu_insts,
                "rocprofv3_avg_us": avg_us}
    with open(tpath, "w") as f:
        json.dump(allt, f, indent=1, sort_keys=True)
    b = os.path.join(out, "bench_%s.json" % wl)
    if os.path.isfile(b) and os.path.getsize(b):
        with open(b) as f, open(os.path.join(here, "%s_%s_bench.json" % (tag, wl)), "w") as g:
            g.write(f.read())
    print(open(os.path.join(here, "%s_%s_kernel_stats.csv" % (tag, wl))).read())
    print(json.dumps(pmc["kernels"], indent=1)[:3000])


if __name__ == "__main__":
    main()
