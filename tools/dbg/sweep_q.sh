#!/bin/bash
for q in 1 8 16 24 32 48; do
  PT_QUORUM=$q bash tools/dbg/ab.sh "quorum$q" atrium260k blob7m 2>&1 | grep quorum
done
for r in 4 8 24; do
  PT_REFILL=$r bash tools/dbg/ab.sh "refill$r" atrium260k blob7m 2>&1 | grep refill
done
