"""Probe, not a test: the pruning walks on rays whose origins lie just inside their range (by default between 3 and 8
times the scene's largest coordinate on the largest axis, which is where their margins are thinnest), aimed at
vertices of the scene with a small scatter and grazing past them: closest hit in both pruning modes and any hit
against the oracle.  usage: python tools/dbg/near_limit.py [rays per scene]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api, scenes
from tests import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
lo_f, hi_f = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (3.0, 7.99)
total = 0
for name, mk in (("box40", lambda: scenes.box_room(40)), ("atrium_small", lambda: scenes.atrium(24, 6, 16, 8)),
                 ("blob_small", lambda: scenes.dense_blob(96)), ("terrain_small", lambda: scenes.terrain_lights(48, 20))):
    meshes, mats = mk()
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    allv = np.concatenate([m[0] for m in meshes])
    M = float(np.abs(allv).max())
    rng = np.random.default_rng(11)
    u = rng.standard_normal((n, 3))
    u /= np.abs(u).max(1, keepdims=True)                      # on the surface of the unit cube
    o = (u * rng.uniform(lo_f, hi_f, (n, 1)) * M).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)] + rng.normal(0, 0.02 * M, (n, 3)) * (rng.random((n, 1)) < 0.5)
    d = (tgt - o).astype(np.float32)
    ref = w.castray(o, d)
    bad = {}
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
        got = ds.trace_closest(o, d, mode)
        bad[mode] = int(((util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"])).sum())
    any_ref = ref["tri"] >= 0
    bad["any"] = int((ds.trace_any(o, d, api.PT_TRACE_FAST) != any_ref).sum())
    total += sum(bad.values())
    print("%s: %d rays from %g-%g x the largest coordinate (%.3g), %d hit; mismatches fast %d, fast-binary %d, any hit %d"
          % (name, n, lo_f, hi_f, M, int(any_ref.sum()), bad[api.PT_TRACE_FAST], bad[api.PT_TRACE_FAST_BINARY], bad["any"]), flush=True)
    ds.close()
print("TOTAL mismatches", total)
