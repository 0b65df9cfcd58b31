"""Prints a compact table of one or more bench.py JSON lines.  usage: bench_summary.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        r = json.load(open(f))
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    ro = r.get("roofline") or {}
    print("%s: %s %.0f Mrays/s %.1f ms" % (f, r["config"]["workload"].split(":")[0], r["value"], r["ms_per_step"]),
          {k: round(v, 1) for k, v in r["kernel_ms_per_step"].items()},
          "closest %.1f boxes %.2f tris | shadow %.1f boxes %.2f tris" % (ro.get("node_boxes_per_ray", 0), ro.get("tri_tests_per_ray", 0),
                                                                 ro.get("shadow_node_boxes_per_ray", 0), ro.get("shadow_tri_tests_per_ray", 0)) if ro else "")
    for o in r.get("other_configs", []):
        print("    %-12s %.0f Mrays/s %.1f ms" % (o["workload"].split(":")[0], o["value"], o["ms_per_step"]),
              {k: round(v, 1) for k, v in o["kernel_ms_per_step"].items()},
              "boxes/ray %.1f tris/ray %.2f rays/sample %.1f" % (o.get("node_boxes_per_ray") or 0, o.get("tri_tests_per_ray") or 0, o.get("rays_per_sample") or 0))
