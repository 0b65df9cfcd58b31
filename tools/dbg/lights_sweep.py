"""Visibility walks with whole vertices per wave against the refill threshold of 16, by number of lights
(the 1 M-triangle terrain under 8 / 16 / 32 / 48 lights, 4K, 4 spp).  usage: python tools/dbg/lights_sweep.py"""
import io, json, os, sys, contextlib
sys.path.insert(0, '.')
from pathtracer_amd import scenes
import bench
for L in [int(x) for x in (sys.argv[1:] or ["8", "16", "32", "48"])]:
    for cohort in ("0", "1"):
        os.environ["PT_COHORT"] = cohort
        scenes.CONFIGS["terrain1m"]["scene"] = (lambda L=L: scenes.terrain_lights(707, L))
        sys.argv = ['bench.py', '--config', 'terrain1m', '--spp', '4', '--steps', '2', '--warmup', '1', '--no-cpu', '--no-others', '--no-count']
        r, w = os.pipe()
        saved = os.dup(1)
        os.dup2(w, 1)
        try:
            bench.main()
        finally:
            os.dup2(saved, 1)
            os.close(w)
        line = os.read(r, 1 << 20).decode().strip().splitlines()[-1]
        os.close(r)
        d = json.loads(line)
        print("lights %d whole-vertex %s: %d Mrays/s, %.1f ms" % (L, cohort, d["value"], d["ms_per_step"]), flush=True)
