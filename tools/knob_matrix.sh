#!/bin/bash
# Runs the GPU parity suite once per tuning-knob setting, so that every alternative code path of
# the kernels (stack spill, split shading, per-lane refill, workgroup sizes, leaf sizes, quorum
# extremes, no LDS node cache) is compared with the oracle, not only the defaults.
for E in "PT_NO_RESIDENT=1" "PT_LEAF=8" "PT_LEAF=1" "PT_BLOCK=256" "PT_BLOCK=512" "PT_STACK_CAP=4" "PT_RES_REFILL=1" \
         "PT_SPLIT=1" "PT_SPLIT=0 PT_NO_RESIDENT=1" "PT_REFILL=1" "PT_REFILL=64" "PT_QUORUM=1 PT_QUORUM_ANY=1" \
         "PT_QUORUM=64 PT_QUORUM_ANY=64" "PT_LDS_BUDGET_KB=1"; do
  echo "== $E"
  env $E timeout -k 10 400 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread \
      -k "not pipelines_agree and not stack_spill" 2>&1 | tail -1
done
