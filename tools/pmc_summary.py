"""Summarise a rocprofv3 counter_collection.csv: median per kernel per counter."""
import csv, sys, collections, statistics, glob
for f in sys.argv[1:]:
    for path in glob.glob(f, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "odinn" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(k, {c: round(statistics.median(v), 1) for c, v in cs.items()}, "n=", len(next(iter(cs.values()))))
