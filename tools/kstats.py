"""Print the top rows of a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir-or-file> [n]"""
import csv, glob, os, re, sys
p = sys.argv[1]
f = p if os.path.isfile(p) else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("odinn::", "")
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", name)
    print("%-70s %6s calls %9.2f ms total %9.1f us avg %5s %%" % (name[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
