"""Turns a counter summary (tools/pmc_summary.py) into profiles/pmc_<config>.json: HBM-side bytes
per launch of the closest-hit kernel.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte read requests of this kernel's coalesced queue reads at 64 bytes, so it is
doubled (checked against the known 24 bytes per ray the kernel reads from the ray queue).
Also: vector instructions per launch of that kernel (SQ_INSTS_VALU) and, when a second summary with
SQ_THREAD_CYCLES_VALU is given, the mean number of active lanes per vector instruction
(SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; 64 = all lanes).
usage: pmc_to_traffic.py <pmc_summary.json> <config> [<pmc_lanes.json>] > profiles/pmc_<config>.json"""
import json, sys

s = json.load(open(sys.argv[1]))
out = {"config": sys.argv[2], "source": sys.argv[1], "kernels": {}}
for name, c in s.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    n = max(1, int(c["_dispatches"]))
    out["kernels"][name] = {
        "dispatches": n,
        "fetch_bytes_per_launch": 2.0 * c["FETCH_SIZE"] * 1024.0 / n,
        "write_bytes_per_launch": c["WRITE_SIZE"] * 1024.0 / n,
    }
    if "k_trace_queue" in name and "false" in name.split("<")[1].split(",")[2]:
        k = out["kernels"][name]
        out["trace_closest_hbm_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
        if "SQ_INSTS_VALU" in c:
            out["trace_closest_valu_insts_per_launch"] = c["SQ_INSTS_VALU"] / n
if len(sys.argv) > 3:
    for name, c in json.load(open(sys.argv[3])).items():
        if "k_trace_queue" in name and "false" in name.split("<")[1].split(",")[2] and c.get("SQ_ACTIVE_INST_VALU"):
            out["trace_closest_active_lanes"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
json.dump(out, sys.stdout, indent=1, sort_keys=True)
