#!/bin/bash
for q in 16 24 40; do
  PT_QUORUM=$q bash tools/dbg/ab.sh "quorum$q" atrium260k blob7m 2>&1 | grep quorum
done
for r in 8 24; do
  PT_REFILL=$r bash tools/dbg/ab.sh "refill$r" atrium260k blob7m 2>&1 | grep refill
done
for q in 16 24 48; do
  PT_QUORUM_ANY=$q bash tools/dbg/ab.sh "qany$q" atrium260k terrain1m 2>&1 | grep qany
done
