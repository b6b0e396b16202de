"""median of one PMC counter over the dispatches of one kernel: pmc_one.py <dir> <kernel-substring> <counter> [tag]"""
import csv, glob, statistics, sys
d, pat, c = sys.argv[1:4]
tag = sys.argv[4] if len(sys.argv) > 4 else ""
v = [float(r["Counter_Value"]) for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(p))
     if pat in r["Kernel_Name"] and r["Counter_Name"] == c]
m = statistics.median(v) if v else None
print(tag, c, "median KiB", m, "-> B/cell at 8 x 1024^2:", (m * 1024 * (2 if c == "FETCH_SIZE" else 1) / (8 * 1024 * 1024)) if v else None)
