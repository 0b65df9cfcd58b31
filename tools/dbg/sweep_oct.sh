#!/bin/bash
# Refill threshold, leaf quorums and the order of visibility items on the eight-child kernels.
mkdir -p gpurun_out
run() {  # label, configs...
  L=$1; shift
  bash tools/dbg/ab.sh "$L" "$@" 2>&1 | grep "^$L" | cut -c1-200 | tee -a gpurun_out/sweep_oct.txt
}
run base atrium260k blob7m terrain1m || exit 1
for r in 4 8 12 24; do PT_REFILL=$r run "refill$r" atrium260k blob7m terrain1m || exit 1; done
for q in 16 24 40 48; do PT_QUORUM=$q run "quorum$q" atrium260k blob7m || exit 1; done
for q in 16 24 40 48; do PT_QUORUM_ANY=$q run "qany$q" atrium260k terrain1m || exit 1; done
PT_SHADOW_ORDER=vertex run vmajor atrium260k blob7m terrain1m
