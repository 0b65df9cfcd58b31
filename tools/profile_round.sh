#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the default benchmark command, then the
# counter passes (each in a run of its own), all under gpurun_out/<tag>/.
#   tools/profile_round.sh r01
set -e
TAG=${1:-r01}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
SHORT="--steps 1 --warmup 0 --no-cpu --no-count --no-others"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o box1k -- python3 bench.py --no-others > $O/bench_under_rocprof.json 2> $O/trace.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_write.log
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_sq.log
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $O/pmc_sq2 -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_sq2.log
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq3 -o p -- python3 bench.py $SHORT > /dev/null 2> $O/pmc_sq3.log
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 > $O/pmc_summary.json
python3 tools/pmc_summary.py $O/pmc_sq3 > $O/pmc_lanes.json
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.log
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq2 $O/pmc_sq3
echo done
