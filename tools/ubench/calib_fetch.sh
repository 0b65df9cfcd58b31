#!/bin/bash
# Runs on the GPU box.  Calibrates the memory-side read counters of L2 on scattered 16-byte-per-lane gathers: the
# chain benchmark reads a known number of records (blocks*256*(50+2000)) from a table far larger than L2 and the
# Infinity Cache, so nearly every record is one line from HBM.  Prints the raw request counters per record.
set -e
O=gpurun_out/calib_fetch
mkdir -p $O
export TMPDIR=/tmp
for rec in 128 64; do
  for mb in 2000 8; do
    timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $O/r${rec}_${mb} -o p -- tools/ubench/gather_chain $mb $rec 4 > $O/r${rec}_${mb}.txt 2>&1
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f${rec}_${mb} -o p -- tools/ubench/gather_chain $mb $rec 4 > $O/f${rec}_${mb}.txt 2>&1
    timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/h${rec}_${mb} -o p -- tools/ubench/gather_chain $mb $rec 4 > $O/h${rec}_${mb}.txt 2>&1
  done
done
python3 - <<'P'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/calib_fetch/*/")):
    tot = collections.defaultdict(float)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "chain" in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    recs = 1024 * 256 * 2050.0
    print(d.split("/")[-2], {k: round(v / recs, 4) for k, v in tot.items()}, "per record")
P
