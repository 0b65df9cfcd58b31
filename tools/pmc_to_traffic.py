"""Turns a counter summary (tools/pmc_summary.py) into profiles/pmc_<config>.json: per launch of the
closest-hit kernel (k_trace_queue, the plain instantiation without work counters), the HBM-side bytes,
the L2 hit rate, the wave-level vector instructions, the share of LDS cycles lost to bank conflicts and
the mean number of active lanes per vector instruction; plus fetch / write bytes of every kernel.
HBM-side read bytes come from the request counters of L2's memory side, sizes resolved: 32 bytes for every
TCC_EA0_RDREQ_32B request, 128 bytes (one line) for every other TCC_EA0_RDREQ request.  That is what
profiles/r03_fetch_calibration.txt measured on this chip for scattered 16-byte-per-lane gathers and for wide
streaming reads alike: one request per missed line, none of them 32-byte, TCC_BUBBLE = 0, and FETCH_SIZE
(which tallies 64 bytes per request on gfx950) exactly half of the bytes read.  Without the request counters
the doubled FETCH_SIZE is used (the same number whenever no 32-byte request occurs).  Infinity-Cache hits are
on the memory side of L2 and are counted.  WRITE_SIZE is in KiB and exact for 16-byte-per-lane stores.
usage: pmc_to_traffic.py <pmc_summary.json> <config> [<pmc_lanes.json> [<spp of the counter runs>]] > profiles/pmc_<config>.json"""
import json, sys


def is_plain(name, kernel):
    if kernel + "<" not in name:
        return False
    args = name.split("<", 1)[1].split(">")[0].split(",")
    return len(args) >= 3 and args[2].strip() == "false"   # MODE, BLOCK, COUNT, ...


def is_plain_trace(name):
    return is_plain(name, "k_trace_queue")


s = json.load(open(sys.argv[1]))
out = {"config": sys.argv[2], "source": sys.argv[1], "kernels": {}}
for name, c in s.items():
    n = max(1, int(c["_dispatches"]))
    k = {"dispatches": n}
    if "TCC_EA0_RDREQ_sum" in c:
        n32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        k["fetch_bytes_per_launch"] = (32.0 * n32 + 128.0 * (c["TCC_EA0_RDREQ_sum"] - n32)) / n
        k["read_requests_per_launch"] = c["TCC_EA0_RDREQ_sum"] / n
        k["read_requests_32B_per_launch"] = n32 / n
        if "FETCH_SIZE" in c:
            k["fetch_size_counter_bytes_per_launch"] = c["FETCH_SIZE"] * 1024.0 / n
    elif "FETCH_SIZE" in c:
        k["fetch_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024.0 / n
    if "WRITE_SIZE" in c:
        k["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024.0 / n
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_INSTS_VALU" in c:
        k["valu_insts_per_launch"] = c["SQ_INSTS_VALU"] / n
        k["salu_insts_per_launch"] = c.get("SQ_INSTS_SALU", 0) / n
    if c.get("SQ_LDS_IDX_ACTIVE"):
        k["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVE_CYCLES"):
        k["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        k["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if len(k) > 1 and not name.startswith("void at::") and "rocclr" not in name:
        out["kernels"][name] = k
    if is_plain_trace(name):
        if "fetch_bytes_per_launch" in k and "write_bytes_per_launch" in k:
            out["trace_closest_hbm_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
            out["hbm_bytes_how"] = ("reads: 128 B x TCC_EA0_RDREQ (32 B for the TCC_EA0_RDREQ_32B ones), writes: WRITE_SIZE"
                                    if "read_requests_per_launch" in k else "reads: 2 x FETCH_SIZE, writes: WRITE_SIZE")
        for src, dst in (("l2_hit_rate", "trace_closest_l2_hit_rate"), ("valu_insts_per_launch", "trace_closest_valu_insts_per_launch"),
                         ("lds_conflict_frac", "trace_closest_lds_conflict_frac"), ("wait_any_frac", "trace_closest_wait_any_frac")):
            if src in k:
                out[dst] = k[src]
    if is_plain(name, "k_shadow_walk"):  # the visibility walk of split shading, the same figures
        if "fetch_bytes_per_launch" in k and "write_bytes_per_launch" in k:
            out["visibility_hbm_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
        for src in ("l2_hit_rate", "valu_insts_per_launch", "lds_conflict_frac", "wait_any_frac"):
            if src in k:
                out["visibility_" + src] = k[src]
if len(sys.argv) > 3:
    for name, c in json.load(open(sys.argv[3])).items():
        if is_plain_trace(name) and c.get("SQ_ACTIVE_INST_VALU"):
            out["trace_closest_active_lanes"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        if is_plain(name, "k_shadow_walk") and c.get("SQ_ACTIVE_INST_VALU"):
            out["visibility_active_lanes"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
if len(sys.argv) > 4:
    out["spp"] = int(sys.argv[4])  # bench.py applies per-launch figures only to runs of the same sample count
json.dump(out, sys.stdout, indent=1, sort_keys=True)
