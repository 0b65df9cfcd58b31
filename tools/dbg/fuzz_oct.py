"""Probe, not a test: the eight-child tree against the oracle on seeded triangle soups without ill-conditioned
triangles (those scenes keep the four-child tree): closest hit and any hit in fast mode for random rays, rays aimed at
vertices, two generations of surface rays with long and short direction vectors, plus a small render.  Run with
PT_OCT_ALWAYS=1 PT_NO_RESIDENT=1 so that small soups use that tree too.
   python tools/dbg/fuzz_oct.py FIRST LAST"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pt_oracle as O
from pathtracer_amd import api
from tests import util

first, last = int(sys.argv[1]), int(sys.argv[2])
total = used = 0
for seed in range(first, last + 1):
    t0 = time.time()
    rng = np.random.default_rng(5000 + seed)
    n_mesh = int(rng.integers(1, 7))
    scale = float(10.0 ** rng.integers(-2, 3))
    meshes, mats = [], []
    for m in range(n_mesh):
        nt = int(rng.integers(50, 3000))
        kind = rng.integers(0, 3)
        c = rng.uniform(-1, 1, (nt, 1, 3))
        if kind == 1:
            c[:, :, int(rng.integers(0, 3))] = rng.choice([-1.0, 0.0, 1.0])       # triangles about a coordinate plane
        v = (c + rng.normal(0, rng.uniform(0.01, 0.3), (nt, 3, 3)))
        if kind == 2:
            ax = int(rng.integers(0, 3))
            v[:, :, ax] = np.round(c[:, :, ax] * 4) / 4                          # exactly flat, axis-aligned triangles (flat gates)
        v = np.ascontiguousarray((v * scale).reshape(-1, 3), np.float32)
        meshes.append((v, np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3), m))
        mats.append(((0.5, 0.5, 0.5), (2.0, 2.0, 2.0) if m == 0 else (0, 0, 0)))
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    if ds.info["n_oct_nodes"] == 0:
        print("seed %d: no eight-child tree (ill-conditioned triangle), skipped" % seed, flush=True)
        ds.close()
        continue
    used += 1
    allv = np.concatenate([m[0] for m in meshes])
    n = 30000
    o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(np.float32)
    d = (rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-2, 2, (n, 1))).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)]
    ax = rng.standard_normal((n, 3)).astype(np.float32)
    ax[np.arange(n), rng.integers(0, 3, n)] = 0                                   # rays parallel to a coordinate plane
    sets = {"random": (o, d), "at_vertices": (o, (tgt - o).astype(np.float32)), "axis": (o, ax)}
    so, sd = sets["at_vertices"]
    for gen in range(2):
        ref = w.castray(np.ascontiguousarray(so), np.ascontiguousarray(sd))
        hit = ref["tri"] >= 0
        if hit.sum() < 10: break
        p = (so[hit] + sd[hit] * ref["t"][hit, None]).astype(np.float32)
        nd = (rng.standard_normal(p.shape) * 10.0 ** rng.uniform(-2, 1, (len(p), 1))).astype(np.float32)
        nd[rng.random(len(p)) < 0.3, int(rng.integers(0, 3))] *= 1e-4             # nearly parallel to a coordinate plane
        sets["surface%d" % gen] = (p, nd)
        so, sd = p, nd
    bad_seed = 0
    for label, (ro, rd) in sets.items():
        ro = np.ascontiguousarray(ro, np.float32); rd = np.ascontiguousarray(rd, np.float32)
        ref = w.castray(ro, rd)
        got = ds.trace_closest(ro, rd, api.PT_TRACE_FAST)
        hit = ref["tri"] >= 0
        bad = (util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"]) | \
              (((util.bits(ref["u"]) != util.bits(got["u"])) | (util.bits(ref["v"]) != util.bits(got["v"]))) & hit)
        bad_any = hit != ds.trace_any(ro, rd, api.PT_TRACE_FAST)
        if bad.any() or bad_any.any():
            bad_seed += int(bad.sum()) + int(bad_any.sum())
            i = int(np.nonzero(bad | bad_any)[0][0])
            print("  seed %d %s: closest %d any %d of %d; first ray %d o %s d %s ref (t %r tri %d) got (t %r tri %d)" % (
                seed, label, bad.sum(), bad_any.sum(), len(bad), i, ro[i].tolist(), rd[i].tolist(),
                float(ref["t"][i]), ref["tri"][i], float(got["t"][i]), got["tri"][i]), flush=True)
    cam_kw = dict(origin=tuple(float(x) for x in rng.uniform(-0.5, 0.5, 3) * scale), euler=tuple(float(x) for x in rng.uniform(-3, 3, 3)), vw=16, vh=9)
    ref_img, _ = w.render(O.default_camera(**cam_kw), 64, 36, 2, seed=seed, rng=0, form=1)
    img, _ = ds.render(api.make_camera(**cam_kw), 64, 36, 2, seed=seed)
    fin = np.isfinite(ref_img)
    px = int(((util.bits(ref_img) != util.bits(img)) & fin).any(-1).sum()) + int((np.isfinite(img) != fin).any(-1).sum())
    bad_seed += px
    ds.close()
    total += bad_seed
    print("seed %d: %d meshes, %d triangles, scale %g, %d ray sets + render, mismatches %d (%.1f s)" % (
        seed, len(meshes), len(allv) // 3, scale, len(sets), bad_seed, time.time() - t0), flush=True)
print("TOTAL mismatches", total, "over", used, "scenes with the eight-child tree")
