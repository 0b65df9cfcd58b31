"""Per-launch kernel durations of a rocprofv3 --kernel-trace CSV, in launch order, one row per depth level:
   tools/trace_passes.py kernel_trace.csv"""
import csv, sys, subprocess, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = sorted({r["Kernel_Name"] for r in rows})
dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")))
def short(n):
    d = dem.get(n, n)
    d = d.split("(")[0].split("<")[0].split()[-1]
    return d
t0 = int(rows[0]["Start_Timestamp"])
tot = collections.Counter()
line = []
for r in rows:
    k = short(r["Kernel_Name"]); dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[k] += dur
    if k == "k_trace_queue" and line:
        print(" ".join(line)); line = []
    line.append("%s %.0fus@%.2fms" % (k.replace("k_", ""), dur, (int(r["Start_Timestamp"]) - t0) / 1e6))
print(" ".join(line))
print({k: round(v / 1e3, 2) for k, v in tot.items()})
print("span ms", (int(rows[-1]["End_Timestamp"]) - t0) / 1e6)
