"""Summarise rocprofv3 --pmc CSVs (one directory per pass) per kernel-name substring."""
import csv, glob, sys, collections, json
root, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); ndisp = collections.defaultdict(int)
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); ndisp[r["Counter_Name"]] += 1
out = {k: {"sum": v, "dispatches": ndisp[k], "per_dispatch": v / max(1, ndisp[k])} for k, v in sorted(tot.items())}
print(json.dumps(out, indent=1))
