#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the benchmark command of one configuration, then the
# counter passes (each in a run of its own: rocprofv3 does not split counters over passes), all under
# gpurun_out/<tag>_<config>/.  The summaries worth keeping are copied to profiles/ by hand afterwards.
#   tools/profile_round.sh r02 box1k [spp-of-the-counter-passes]
set -e
TAG=${1:-r02}
CFG=${2:-box1k}
SPP=${3:-0}
ROOT=$(pwd)
O=$ROOT/gpurun_out/${TAG}_${CFG}
mkdir -p $O
export TMPDIR=/tmp
SHORT="--config $CFG --steps 1 --warmup 0 --no-cpu --no-count --no-others"
if [ "$SPP" != "0" ]; then SHORT="$SHORT --spp $SPP"; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py $SHORT --steps 2 --warmup 1 > $O/bench_under_rocprof.json 2> $O/trace.log
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_$name.log
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM
pass sq3 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_ea $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.json
python3 tools/pmc_summary.py $O/pmc_sq3 > $O/pmc_lanes.json
USED_SPP=$(python3 -c "import json,re;print(int(re.search(r'(\d+) spp per GPU', json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1])['config']['workload']).group(1)))")
python3 tools/pmc_to_traffic.py $O/pmc_summary.json $CFG $O/pmc_lanes.json $USED_SPP > $O/pmc_$CFG.json
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/trace -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_ea $O/pmc_sq $O/pmc_sq2 $O/pmc_sq3 $O/trace
echo done $CFG
