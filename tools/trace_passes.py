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
# time with no kernel running between the first start and the last end, and its share after k_shade / k_tail
# launches (the places where the host waits for a count before it launches again)
idle = 0.0; after_sync = 0.0; end = int(rows[0]["End_Timestamp"])
for prev, r in zip(rows, rows[1:]):
    gap = int(r["Start_Timestamp"]) - end
    if gap > 0:
        idle += gap
        if short(prev["Kernel_Name"]) in ("k_shade", "k_tail", "k_resolve"): after_sync += gap
    end = max(end, int(r["End_Timestamp"]))
print("idle ms %.3f, of which after a kernel the host waits for %.3f" % (idle / 1e6, after_sync / 1e6))
