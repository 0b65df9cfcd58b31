"""The named benchmark configurations at their REAL scene sizes, against the oracle.

  config 3  atrium260k   259 562 triangles, 18 meshes, 1 light; 5 bounces, thin lens on
  config 4  blob7m       6 998 542 triangles (the 7 M-triangle blob, deep tree, inside the 1 002-triangle room), 1 light
  config 5  terrain1m    999 826 triangles, 65 meshes, 64 area lights (a real top-level tree)

For each: closest hits (t, u, v bit patterns + triangle id) and any-hit answers of the HIP walk equal
the oracle's recursive BVH::intersect (object.cpp:265-285, 406-429) on primary, random, axis-aligned
and surface-origin rays; a reduced frame (160x90) of the full scene equals the oracle's render bit
for bit; the named frame size (1080p / 4K) is checked through properties that need no CPU render
of that size; fast and exact traversal agree on millions of rays with 0 mismatches.
The scene is built twice, by the product's host builder and by the oracle, and both trees are walked."""
import numpy as np
import pytest

from tests import util
from oracle import pt_oracle as O

pytestmark = pytest.mark.gpu

NAMES = ["atrium260k", "blob7m", "terrain1m"]
# ray-origin boxes of the random ray sets; camera, integrator settings and frame size are the configuration's own
BOUNDS = {"atrium260k": ((-2, -2, -1), (2, 2, 5)), "blob7m": ((-2, -2, -1), (2, 2, 5)), "terrain1m": ((-9, -3, -1), (9, 3, 17))}


def _setup(name):
    from pathtracer_amd import scenes
    c = scenes.CONFIGS[name]
    cam = {k: v for k, v in c["cam"].items() if k != "fov"}
    return dict(cam=cam, opt=dict(c.get("opt", {})), W=c["W"], H=c["H"], lo=BOUNDS[name][0], hi=BOUNDS[name][1])


SETUP = {n: _setup(n) for n in NAMES}

@pytest.fixture(scope="module")
def api():
    from pathtracer_amd import api
    assert api.device_count() >= 1
    return api


@pytest.fixture(scope="module", params=NAMES)
def cfg(request):
    """(name, oracle world, host scene, device scene) of a named configuration.  Module-scoped and
    parametrized, so pytest runs all tests of one scene before it builds the next (they are large)."""
    from pathtracer_amd import host, scenes
    name = request.param
    meshes, mats = scenes.CONFIGS[name]["scene"]()
    w = O.World.from_meshes(meshes, mats)
    hs = host.Scene.from_meshes(meshes, mats)
    ds = hs.device_scene(0)
    assert ds.info["n_tris"] == w.n_tris == scenes.count_tris(meshes)
    del meshes
    yield name, w, hs, ds
    hs.close()


def check_equal(ref, got, label):
    bad_t = util.bits(ref["t"]) != util.bits(got["t"])
    bad_i = ref["tri"] != got["tri"]
    hit = ref["tri"] >= 0
    bad_u = (util.bits(ref["u"]) != util.bits(got["u"])) & hit
    bad_v = (util.bits(ref["v"]) != util.bits(got["v"])) & hit
    n_bad = int((bad_t | bad_i | bad_u | bad_v).sum())
    assert n_bad == 0, "%s: %d of %d rays differ (t %d, tri %d, u %d, v %d)" % (
        label, n_bad, len(bad_t), bad_t.sum(), bad_i.sum(), bad_u.sum(), bad_v.sum())


def test_closest_hits_on_the_full_scene_are_bit_exact(api, cfg):
    """>= 20 000 primary + 20 000 random + axis-aligned + surface-origin rays in fast mode, a subset in
    exact mode (which walks what the reference walks: most of the tree for an axis-aligned ray)."""
    name, w, hs, ds = cfg
    S = SETUP[name]
    pin = {k: v for k, v in S["cam"].items() if k not in ("aperture", "focus")}
    sets = {
        "primary": util.primary_rays(O.default_camera(**pin), 192, 108),
        "primary_tilted": util.primary_rays(O.default_camera(origin=(0.2, 0.4, 0.1), euler=(0.1, 0.3, 0.05)), 128, 128),
        "random": util.random_rays(20000, 5, S["lo"], S["hi"]),
        "axis": util.axis_rays(4000, 6, S["lo"], S["hi"]),
    }
    sets["surface"] = util.surface_rays(w, 20000, 7, cam=O.default_camera(**pin))
    for label, (o, d) in sets.items():
        ref = w.castray(o, d)
        assert (ref["tri"] >= 0).sum() > 0 or label == "axis"
        check_equal(ref, ds.trace_closest(o, d, api.PT_TRACE_FAST), "%s/%s/fast" % (name, label))
        assert ((ref["tri"] >= 0) == ds.trace_any(o, d, api.PT_TRACE_FAST)).all(), "%s/%s any-hit" % (name, label)
        if label in ("primary_tilted", "random", "surface"):
            k = slice(0, 6000)
            sub = {key: val[k] for key, val in ref.items()}
            check_equal(sub, ds.trace_closest(o[k], d[k], api.PT_TRACE_EXACT), "%s/%s/exact" % (name, label))


def test_reduced_frame_of_the_full_scene_matches_the_oracle_bitwise(api, cfg):
    """160x90, 2 spp, with the configuration's own integrator settings (5 bounces + thin lens for the
    260 k scene; 64 lights sampled at every vertex for the terrain): radiance sums equal bit for bit
    on all but a stated 0.1 % of pixels (double-precision libm results that round to another float),
    ray counts within the same bound."""
    name, w, hs, ds = cfg
    S = SETUP[name]
    W, H, spp = 160, 90, 2
    ref, rc = w.render(O.default_camera(**S["cam"]), W, H, spp, seed=17, opt=O.default_options(**S["opt"]), rng=0, form=1)
    got, gc = ds.render(api.make_camera(**S["cam"]), W, H, spp, seed=17, opt=api.make_options(**S["opt"]))
    assert np.isfinite(got).all() == np.isfinite(ref).all()
    same = (util.bits(ref) == util.bits(got)).all(-1)
    print("%s: %d of %d pixels differ from the oracle" % (name, (~same).sum(), same.size))  # shown with -s / -rP
    assert same.all(), "%s: %d of %d pixels differ" % (name, (~same).sum(), same.size)
    assert int(gc.rays_closest) == int(rc.rays_closest) and int(gc.rays_shadow) == int(rc.rays_shadow)
    assert gc.paths == rc.paths == W * H * spp
    assert got.sum() > 0
    if S["opt"].get("max_depth"):
        assert gc.bounce_passes == S["opt"]["max_depth"] + 1   # depths 0..k are traced, then the path ends


def test_named_frame_size_properties(api, cfg):
    """1080p (4K for the terrain) with the configuration's settings, through properties that need no CPU
    render of that size: one call == the same samples split over two calls, bit for bit; the strided
    shards two GPUs would render add up to it; ray counts are additive; nothing is non-finite."""
    name, w, hs, ds = cfg
    S = SETUP[name]
    W, H = S["W"], S["H"]
    cam = api.make_camera(**S["cam"])
    one, c1 = ds.render(cam, W, H, 2, seed=8, opt=api.make_options(**S["opt"]))
    two, _ = ds.render(cam, W, H, 1, seed=8, opt=api.make_options(**S["opt"]))
    two, _ = ds.render(cam, W, H, 1, spp_begin=1, seed=8, opt=api.make_options(**S["opt"]), accum=two)
    assert (util.bits(one) == util.bits(two)).all()
    s0, c0 = ds.render(cam, W, H, 1, spp_begin=0, seed=8, opt=api.make_options(spp_stride=2, **S["opt"]))
    s1, c2 = ds.render(cam, W, H, 1, spp_begin=1, seed=8, opt=api.make_options(spp_stride=2, **S["opt"]))
    np.testing.assert_allclose(s0 + s1, one, rtol=2e-5, atol=1e-6)
    assert c0.rays_closest + c2.rays_closest == c1.rays_closest
    assert c0.rays_shadow + c2.rays_shadow == c1.rays_shadow
    assert np.isfinite(one).all() and c1.paths == W * H * 2
    assert (one.sum(-1) > 0).mean() > 0.05


def _fast_vs_exact(api, name, ds, lo, hi):
    n = 1_500_000
    rng = np.random.default_rng(len(name))
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[rng.random(n) < 0.02, 0] = 0.0
    d[rng.random(n) < 0.02, 2] = 0.0
    for gen in range(2):
        a = ds.trace_closest(o, d, mode=api.PT_TRACE_EXACT)
        b = ds.trace_closest(o, d, mode=api.PT_TRACE_FAST)
        ha = np.asarray(ds.trace_any(o, d, mode=api.PT_TRACE_EXACT))
        hb = np.asarray(ds.trace_any(o, d, mode=api.PT_TRACE_FAST))
        diff = (a["tri"] != b["tri"]) | (util.bits(a["t"]) != util.bits(b["t"])) | \
               (util.bits(a["u"]) != util.bits(b["u"])) | (util.bits(a["v"]) != util.bits(b["v"]))
        assert int(diff.sum()) == 0, "%s gen %d: %d closest-hit mismatches" % (name, gen, diff.sum())
        assert int((ha != hb).sum()) == 0, "%s gen %d: %d any-hit mismatches" % (name, gen, (ha != hb).sum())
        assert int((ha != (a["tri"] >= 0)).sum()) == 0
        hit = a["tri"] >= 0
        o = (o[hit] + d[hit] * a["t"][hit, None]).astype(np.float32)
        d = rng.standard_normal((len(o), 3)).astype(np.float32)
        if len(o) == 0:
            break


def test_fast_and_exact_traversal_agree_on_millions_of_rays(api, cfg):
    """Fast mode only skips work that cannot change the answer: on 1.5 M random rays (some axis-parallel)
    and on the rays that restart from their hit points (the ill-conditioned surface-origin case every
    bounce produces) both modes return the same t, u, v bits and triangle, the same any-hit answer, and
    any-hit equals (closest hit exists).  0 mismatches."""
    name, w, hs, ds = cfg
    _fast_vs_exact(api, name, ds, np.array(SETUP[name]["lo"]) - 0.5, np.array(SETUP[name]["hi"]) + 0.5)


def test_fast_and_exact_traversal_agree_on_the_room(api):
    from pathtracer_amd import host, scenes
    hs = host.Scene.from_meshes(*scenes.CONFIGS["box1k"]["scene"]())
    _fast_vs_exact(api, "box1k", hs.device_scene(0), (-2.5, -2.5, -1.5), (2.5, 2.5, 5.5))
    hs.close()
