"""print name, calls, average us of the kernels of a rocprofv3 --stats output directory whose names match a pattern"""
import csv, glob, re, sys
d, pat = sys.argv[1], re.compile(sys.argv[2])
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat.search(row["Name"]):
            print("  %-60s calls %5s avg %9.1f us" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
