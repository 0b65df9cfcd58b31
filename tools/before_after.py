"""Writes profiles/r03_before_after.txt: the closest-hit and visibility kernels of the scenes outside LDS in round 2
(four-child tree) and round 3 (eight-child tree), from the counter summaries and bench records of both rounds."""
import json

R02 = {  # records per ray of round 2 (measured then; its bench record carries no per-configuration counts) and its sample counts
    "atrium260k": ("15.2 node records per ray (60.9 child boxes)", 32),
    "blob7m": ("20.0 node + 2.4 leaf-pair records per ray (80 child boxes, 4.7 triangles)", 16),
    "terrain1m": ("15.9 node records per ray (63.7 child boxes)", 2),
}


def plain(summary, kernel):
    for k, v in summary.items():
        if kernel + "<" in k and k.split("<", 1)[1].split(">")[0].split(",")[2].strip() == "false":
            return v
    return None


def bench(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    out = {"box1k": d}
    for c in d.get("configs", d.get("other_configs", [])):
        out[c["workload"].split(":")[0]] = c
    return out


b2, b3 = bench("profiles/r02_bench.json"), bench("profiles/r03_bench.json")
lines = ["# Scenes outside LDS, closest-hit kernel (and the visibility kernel), round 2 (four-child tree, 112-byte nodes) against round 3",
         "# (eight-child tree with boxes on an 8-bit grid, 80-byte nodes in one line; visibility rays vertex by vertex).",
         "# From profiles/r02_* and profiles/r03_* (counter passes at the sample counts of each round's benchmark) and the two bench records.",
         "# records per ray = node fetches (child boxes / children per node) + leaf-pair fetches (triangle tests / 2).", ""]
for c in ("atrium260k", "blob7m", "terrain1m"):
    lines.append(c)
    for tag, b in (("r02", b2), ("r03", b3)):
        s = json.load(open("profiles/%s_%s_pmc_summary.json" % (tag, c)))
        ln = json.load(open("profiles/%s_%s_pmc_lanes.json" % (tag, c)))
        v, vs = plain(s, "k_trace_queue"), plain(s, "k_shadow_walk")
        l2 = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        l2s = vs["TCC_HIT_sum"] / (vs["TCC_HIT_sum"] + vs["TCC_MISS_sum"])
        vl, vsl = plain(ln, "k_trace_queue"), plain(ln, "k_shadow_walk")
        lanes, lanes_s = vl["SQ_THREAD_CYCLES_VALU"] / vl["SQ_ACTIVE_INST_VALU"], vsl["SQ_THREAD_CYCLES_VALU"] / vsl["SQ_ACTIVE_INST_VALU"]
        cfg = b[c]
        if tag == "r02":
            rec, spp = R02[c]
        else:
            rf = cfg.get("roofline_closest_hit") or cfg["roofline"]
            rec = "%.1f node + %.1f leaf-pair records per ray (%.1f child boxes, %.2f triangles)" % (
                rf["node_boxes_per_ray"] / 8, rf["tri_tests_per_ray"] / 2, rf["node_boxes_per_ray"], rf["tri_tests_per_ray"])
            spp = cfg["spp_per_gpu"]
        lines.append("  %s  closest hit: %s; L2 hit rate %.0f %%, %.1f of 64 lanes per vector instruction; visibility kernel: L2 hit rate %.0f %%, "
                     "%.1f lanes; frame %.2f G rays/s at %s spp" % (tag, rec, 100 * l2, lanes, 100 * l2s, lanes_s, cfg["value"] / 1000, spp))
    lines.append("")
open("profiles/r03_before_after.txt", "w").write("\n".join(lines))
print("\n".join(lines))
