#!/bin/bash
# The probes against the oracle in one go (after a change to margins, ranges or the tree builder): soups in all
# modes, soups through the eight-child tree, the scaled structured scenes, soups rendered, and the range sweep.
# Every probe writes its own file under gpurun_out/ and prints a total; stops at the first that fails to run.
mkdir -p gpurun_out
timeout -k 10 900 python -u tools/dbg/fuzz_soup.py 2000 2400 > gpurun_out/camp_soup.txt 2>&1 || exit 1; tail -2 gpurun_out/camp_soup.txt
PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 timeout -k 10 900 python -u tools/dbg/fuzz_oct.py 700 900 > gpurun_out/camp_oct.txt 2>&1 || exit 1; tail -2 gpurun_out/camp_oct.txt
timeout -k 10 600 python -u tools/dbg/fuzz_scaled.py > gpurun_out/camp_scaled.txt 2>&1 || exit 1; tail -2 gpurun_out/camp_scaled.txt
PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 timeout -k 10 600 python -u tools/dbg/fuzz_scaled.py > gpurun_out/camp_scaled_oct.txt 2>&1 || exit 1; tail -2 gpurun_out/camp_scaled_oct.txt
timeout -k 10 900 python -u tools/dbg/fuzz_render.py 1000 1150 > gpurun_out/camp_render.txt 2>&1 || exit 1; tail -2 gpurun_out/camp_render.txt
timeout -k 10 600 python -u tools/dbg/far.py > gpurun_out/camp_far.txt 2>&1 || exit 1; tail -8 gpurun_out/camp_far.txt
timeout -k 10 600 python -u tools/dbg/near_limit.py 1000000 > gpurun_out/camp_near.txt 2>&1 || exit 1; tail -6 gpurun_out/camp_near.txt
