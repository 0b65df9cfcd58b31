"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel: sum of every counter over the
kernel's dispatches plus the dispatch count.  usage: pmc_summary.py <dir> [<dir> ...] > summary.json"""
import csv, glob, json, os, sys, collections

out = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                out[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
res = {}
for k, c in out.items():
    res[k] = {n: v for n, v in c.items()}
    res[k]["_dispatches"] = max(len(disp[(k, n)]) for n in c)
json.dump(res, sys.stdout, indent=1, sort_keys=True)
