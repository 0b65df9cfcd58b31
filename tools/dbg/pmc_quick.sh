#!/bin/bash
# one SQ counter pass of a configuration's frame; prints per-kernel busy / wait fractions.  usage: pmc_quick.sh CONFIG SPP
export TMPDIR=/tmp
O=gpurun_out/pmcq_$1
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $O -o p -- python3 bench.py --config $1 --spp $2 --steps 1 --warmup 0 --no-cpu --no-count --no-others > /dev/null 2> $O/log.txt
python3 tools/pmc_summary.py $O > $O/sum.json
python3 - <<P
import json
d=json.load(open("$O/sum.json"))
for k,c in d.items():
    if "k_trace_queue" in k or "k_shadow_walk" in k or "k_tail" in k:
        wc=c.get("SQ_WAVE_CYCLES",1)
        print(k[:60], "disp", c["_dispatches"], "valu busy %.2f"%(c["SQ_ACTIVE_INST_VALU"]/wc*4), "wait %.2f"%(c["SQ_WAIT_ANY"]/wc), "active any %.2f"%(c["SQ_ACTIVE_INST_ANY"]/wc), "lanes %.1f"%(c["SQ_THREAD_CYCLES_VALU"]/max(1,c["SQ_ACTIVE_INST_VALU"])), "valu/vmem %.1f"%(c["SQ_INSTS_VALU"]/max(1,c["SQ_INSTS_VMEM_RD"])))
P
