"""How much of a rocprofv3 --kernel-trace CSV's busy time has two or more kernels running at once:
   tools/trace_overlap.py kernel_trace.csv"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
print("kernels %d, busy %.2f ms, two or more at once %.2f ms (%.0f %%), span %.2f ms" % (len(rows), busy / 1e6, over / 1e6, 100.0 * over / max(1, busy), (rows[-1][1] - rows[0][0]) / 1e6))
print("queues/streams:", sorted({r[3] for r in rows}))
t0 = rows[0][0]
for s, e, n, q in rows[-60:]:
    print("%9.3f %9.3f %s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, n))
