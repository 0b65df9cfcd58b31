#!/bin/bash
# Runs the GPU parity tests of the walks and the renderer once per tuning-knob setting, so that every
# alternative code path of the kernels (stack spill, split and fused shading on either kind of scene, per-lane
# refill, workgroup sizes, leaf sizes, quorum extremes, no LDS node cache, tail kernel never / always, the
# eight-child tree on every scene or on none, both collapses, workgroups per CU, both orders of the visibility
# rays, short and odd LDS stacks under the eight-child walk) is
# compared with the oracle, not only the defaults.  Stops at the first setting that fails.
#   tools/knob_matrix.sh [test files...]      (default: the small-scene walk and render tests)
TESTS=${@:-tests/test_gpu_trace.py tests/test_gpu_render.py}
mkdir -p gpurun_out
rm -f gpurun_out/knob_matrix.txt
for E in "PT_NO_RESIDENT=1" "PT_LEAF=8" "PT_LEAF=1" "PT_BLOCK=256" "PT_BLOCK=512" "PT_BLOCK=1024 PT_NO_RESIDENT=1" "PT_STACK_CAP=4" \
         "PT_RES_REFILL=1" "PT_SPLIT=1" "PT_SPLIT=0 PT_NO_RESIDENT=1" "PT_REFILL=1" "PT_REFILL=64" \
         "PT_QUORUM=1 PT_QUORUM_ANY=1" "PT_QUORUM=64 PT_QUORUM_ANY=64" "PT_LDS_BUDGET_KB=1" "PT_TAIL=0" \
         "PT_TAIL=1000000000" "PT_TAIL=1000000000 PT_NO_RESIDENT=1" "PT_MARGIN_SHIFT=20" \
         "PT_NO_OCT=1" "PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1" "PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 PT_STACK_CAP=4" "PT_COLLAPSE_GREEDY=1" \
         "PT_LDS_SHARE=5" "PT_LDS_SHARE=7 PT_NO_RESIDENT=1" "PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 PT_TAIL=1000000000" \
         "PT_SHADOW_ORDER=light PT_NO_RESIDENT=1" "PT_SHADOW_ORDER=light PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1" \
         "PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 PT_STACK=5" "PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 PT_STACK=6 PT_REFILL=1" \
         "PT_SHADOW_GROUP=4 PT_NO_RESIDENT=1" "PT_COHORT=1 PT_NO_RESIDENT=1" "PT_REFILL_ANY=3 PT_NO_RESIDENT=1"; do
  echo "== $E"
  env $E timeout -k 10 400 python3 -u -m pytest $TESTS -m gpu -x -q --timeout 120 --timeout-method thread \
      -k "not pipelines_agree and not stack_spill and not tail_kernel_and_wavefront" > gpurun_out/knob.log 2>&1
  rc=$?
  printf "%-58s %s\n" "$E" "$(tail -1 gpurun_out/knob.log)" | tee -a gpurun_out/knob_matrix.txt
  if [ $rc -ne 0 ]; then tail -30 gpurun_out/knob.log; exit 1; fi
done
