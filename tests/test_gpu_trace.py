"""GPU closest-hit / any-hit parity with the oracle's recursive Scene::castray
(scene.cpp:18-20 -> object.cpp:265-285,406-429).  Bit-exact: t, u, v as uint32 bit
patterns and the triangle id, on every ray, in both traversal modes."""
import numpy as np
import pytest

from tests import util
from oracle import pt_oracle as O

pytestmark = pytest.mark.gpu

SCENES = ["plane", "box", "box40", "atrium_small", "blob_small", "terrain_small"]


@pytest.fixture(scope="module")
def api():
    from pathtracer_amd import api
    assert api.device_count() >= 1
    return api


_DEV = {}


def dev_scene(api, name):
    if name not in _DEV:
        _DEV[name] = api.DeviceScene(util.world(name).export())
    return _DEV[name]


def ray_sets(name):
    w = util.world(name)
    sets = {}
    cams = [O.default_camera(), O.default_camera(origin=(0.3, -0.2, 0.5), euler=(0.2, -0.4, 0.1), vw=16, vh=9),
            O.default_camera(origin=(0, 1, -3), euler=(0.25, 0, 0), fov=40.0)]
    for i, c in enumerate(cams):
        sets["primary%d" % i] = util.primary_rays(c, 160, 120)
    lo, hi = ((-9, -3, -1), (9, 3, 17)) if name == "terrain_small" else ((-2, -2, -1), (2, 2, 5))
    sets["random"] = util.random_rays(30000, 1, lo, hi)
    sets["axis"] = util.axis_rays(8000, 2, lo, hi)
    if name != "plane":
        sets["surface"] = util.surface_rays(w, 30000, 3)
    if name == "box":
        sets["lattice"] = util.lattice_rays()
    return sets


def check_equal(ref, got, label):
    bad_t = util.bits(ref["t"]) != util.bits(got["t"])
    bad_i = ref["tri"] != got["tri"]
    hit = ref["tri"] >= 0
    bad_u = (util.bits(ref["u"]) != util.bits(got["u"])) & hit
    bad_v = (util.bits(ref["v"]) != util.bits(got["v"])) & hit
    n_bad = int((bad_t | bad_i | bad_u | bad_v).sum())
    assert n_bad == 0, "%s: %d of %d rays differ (t %d, tri %d, u %d, v %d)" % (
        label, n_bad, len(bad_t), bad_t.sum(), bad_i.sum(), bad_u.sum(), bad_v.sum())


@pytest.mark.parametrize("name", SCENES)
def test_closest_exact_mode_is_bit_exact(api, name):
    w, ds = util.world(name), dev_scene(api, name)
    for label, (o, d) in ray_sets(name).items():
        check_equal(w.castray(o, d), ds.trace_closest(o, d, api.PT_TRACE_EXACT), "%s/%s/exact" % (name, label))


@pytest.mark.parametrize("name", SCENES)
def test_closest_fast_mode_primary_and_random_bit_exact(api, name):
    w, ds = util.world(name), dev_scene(api, name)
    for label, (o, d) in ray_sets(name).items():
        if label == "surface":
            continue
        check_equal(w.castray(o, d), ds.trace_closest(o, d, api.PT_TRACE_FAST), "%s/%s/fast" % (name, label))


@pytest.mark.parametrize("name", [s for s in SCENES if s != "plane"])
def test_closest_fast_mode_surface_rays(api, name):
    """Rays leaving a surface at grazing angles are ill-conditioned: the reference's own t is
    noise there.  The fast mode's margins must still reproduce it on every ray."""
    w, ds = util.world(name), dev_scene(api, name)
    o, d = ray_sets(name)["surface"]
    ref, got = w.castray(o, d), ds.trace_closest(o, d, api.PT_TRACE_FAST)
    bad = (util.bits(ref["t"]) != util.bits(got["t"])) | (ref["tri"] != got["tri"])
    assert not bad.any(), "%s: %d of %d surface rays differ" % (name, bad.sum(), len(bad))


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("mode", [0, 1])
def test_any_hit_matches_castray(api, name, mode):
    w, ds = util.world(name), dev_scene(api, name)
    for label, (o, d) in ray_sets(name).items():
        ref = w.castray(o, d)["tri"] >= 0
        got = ds.trace_any(o, d, mode)
        bad = ref != got
        assert not bad.any(), "%s/%s mode %d: %d of %d differ" % (name, label, mode, bad.sum(), len(bad))


def test_known_answer_rays_plane(api):
    """The rays whose reference results were recorded from a run of the reference (t as bit patterns, tests/golden/plane_kat.json)."""
    ds = dev_scene(api, "plane")
    o = np.array([(1, 3, -1), (5.153648, 3, 0), (0, 0, 0), (0, 1e-6, 0), (0, 1.5e-6, 0), (0, 3, 0), (0, 3, 0),
                  (-6, 3, .5), (-1, 3, .5)], np.float32)
    d = np.array([(0, -1, 0)] * 5 + [(1, 0, 0), (0, 1, 0), (0, -1, 0), (0, -1, 0)], np.float32)
    for mode in (0, 1):
        r = ds.trace_closest(o, d, mode)
        assert list(util.bits(r["t"])) == [0x403fffff, 0x403fffff, 0x7f800000, 0x7f800000, 0x35c9539b, 0x7f800000,
                                            0x7f800000, 0x7f800000, 0x403fffff]
        assert list(r["tri"]) == [0, 0, -1, -1, 0, -1, -1, -1, 1]
        np.testing.assert_allclose(r["v"][0], 0.597019, rtol=1e-5)
        np.testing.assert_allclose([r["u"][8], r["v"][8]], [0.402981, 0.0485093], rtol=1e-5)


def test_camera_rays_bit_exact(api):
    for cam_o, cam_g in [(O.default_camera(), api.make_camera()),
                         (O.default_camera(euler=(0.3, -0.7, 0.2), fov=47.0, vw=16, vh=9),
                          api.make_camera(euler=(0.3, -0.7, 0.2), fov=47.0, vw=16, vh=9))]:
        a = O.camera_rays(cam_o, 320, 180)
        b = api.camera_rays(cam_g, 320, 180)
        assert (util.bits(a) == util.bits(b)).all()


def test_philox_matches(api):
    for ctr, key in [([0] * 4, [0] * 2), ([0xffffffff] * 4, [0xffffffff] * 2),
                     ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])]:
        assert (api.philox(ctr, key) == O.philox(ctr, key)).all()


def test_deep_stack_spills_to_global(api, monkeypatch):
    """A tree deeper than the LDS part of the traversal stack keeps the rest in global memory."""
    monkeypatch.setenv("PT_STACK_CAP", "4")
    w = util.world("atrium_small")
    ds = api.DeviceScene(w.export())
    assert ds.info["max_depth"] > 4
    for label in ("primary1", "random", "surface"):
        o, d = ray_sets("atrium_small")[label]
        check_equal(w.castray(o, d), ds.trace_closest(o, d, api.PT_TRACE_EXACT), "spill/%s" % label)
        assert ((w.castray(o, d)["tri"] >= 0) == ds.trace_any(o, d, api.PT_TRACE_EXACT)).all()
    ds.close()


@pytest.mark.parametrize("name,builder", [
    ("atrium260k", lambda: __import__("pathtracer_amd.scenes", fromlist=["x"]).atrium(150)),
    ("blob1m", lambda: __import__("pathtracer_amd.scenes", fromlist=["x"]).dense_blob(707)),
])
def test_large_scenes_bit_exact_through_the_host_builder(api, name, builder):
    """Benchmark-sized trees (260 k and 1 M triangles), built by the product's host-side builder, walked on
    the GPU, against the oracle's own build and recursive walk: primary, random and axis-aligned rays."""
    from pathtracer_amd import host
    meshes, mats = builder()
    w = O.World.from_meshes(meshes, mats)
    ds = host.Scene.from_meshes(meshes, mats).device_scene(0)
    assert ds.info["n_tris"] == w.n_tris
    sets = {
        "primary": util.primary_rays(O.default_camera(vw=16, vh=9), 192, 108),
        "primary_tilted": util.primary_rays(O.default_camera(origin=(0.2, 0.1, 0.4), euler=(0.1, 0.3, 0.05)), 128, 128),
        "random": util.random_rays(20000, 5),
        "axis": util.axis_rays(4000, 6),
    }
    for label, (o, d) in sets.items():
        ref = w.castray(o, d)
        check_equal(ref, ds.trace_closest(o, d, api.PT_TRACE_FAST), "%s/%s/fast" % (name, label))
        if label != "axis" and label != "primary":   # exact mode walks most of the tree for axis-aligned rays
            check_equal(ref, ds.trace_closest(o, d, api.PT_TRACE_EXACT), "%s/%s/exact" % (name, label))
        assert ((ref["tri"] >= 0) == ds.trace_any(o, d, api.PT_TRACE_FAST)).all()


@pytest.mark.parametrize("name", ["plane", "box", "atrium_small", "terrain_small", "blob_small"])
@pytest.mark.parametrize("mode", [0, 1])
def test_hits_counter_matches_oracle(name, mode):
    """HitData::hits (object.cpp:203,281-282) for hits and for misses, on scenes whose top level is
    one leaf (box), a leaf of many meshes (atrium) and a real tree (terrain: 21 meshes)."""
    from pathtracer_amd import api
    w = util.world(name)
    ds = api.DeviceScene(w.export())
    cam = O.default_camera(vw=16, vh=9)
    sets = [util.primary_rays(cam, 96, 54), util.random_rays(20000, 21, lo=(-8, -2, -1), hi=(8, 2, 16)),
            util.axis_rays(4000, 22)]
    for o, d in sets:
        ref = w.castray(o, d, want_hits=True)
        got = ds.trace_hits(o, d, mode=mode)
        assert (got == ref["hits"]).all(), "%d of %d differ" % ((got != ref["hits"]).sum(), len(got))
    ds.close()


def test_hits_counter_matches_the_reference_probe():
    """A recorded run of the reference on its 1 002-triangle room: hits of the primary rays that hit
    had mean 5.2 and maximum 7."""
    import json, os
    from pathtracer_amd import api
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "plane_kat.json")))
    mean, mx = gold["box_1002_tris_hits_mean_max"]
    w = util.world("box")
    ds = api.DeviceScene(w.export())
    o, d = util.primary_rays(O.default_camera(), 256, 256)
    hits = ds.trace_hits(o, d)
    tri = ds.trace_closest(o, d)["tri"]
    ds.close()
    assert abs(hits[tri >= 0].mean() - mean) < 0.06 and hits.max() == mx


def _awkward_scene():
    """Geometry a loader can hand over and the reference accepts: two meshes with coincident triangles (exact
    ties between meshes), zero-area triangles (repeated and collinear vertices), slivers, a triangle a million
    units across next to ones a millionth across, triangles whose edges lie on the coordinate axes."""
    from pathtracer_amd import scenes
    gv, gt = scenes._grid_quad((-1.5, -1.5, 3.0), (3, 0, 0), (0, 3, 0), 6)
    odd_v = np.array([
        [0, 0, 2], [0, 0, 2], [0, 0, 2],                      # a point
        [-1, 0, 2.5], [0, 0, 2.5], [1, 0, 2.5],               # collinear
        [-1, 1, 2], [1, 1, 2], [0, 1 + 1e-6, 2],              # sliver
        [-1e6, -1e6, 40], [1e6, -1e6, 40], [0, 1e6, 40],      # huge
        [1e-6, 0, 1], [0, 1e-6, 1], [-1e-6, -1e-6, 1],        # tiny, on the view axis
        [0, 0, 1.5], [1, 0, 1.5], [0, 1, 1.5],                # edges on the axes
        [0, 0, 1.5], [0, -1, 1.5], [-1, 0, 1.5],
    ], np.float32)
    odd_t = np.arange(len(odd_v), dtype=np.uint32).reshape(-1, 3)
    meshes = [(gv, gt, 0), (gv.copy(), gt.copy(), 1), (odd_v, odd_t, 2)]
    mats = [((0.8, 0.8, 0.8), (0, 0, 0)), ((0.2, 0.8, 0.2), (0, 0, 0)), ((0.5, 0.5, 0.5), (1, 1, 1))]
    return meshes, mats


def test_awkward_geometry_is_bit_exact(api):
    """Coincident meshes, zero-area triangles, slivers, extreme sizes, axis-aligned edges: rays through
    vertices and along edges of the grid (ties between neighbours and between the two coincident meshes),
    straight down the axes and random, in every traversal mode, through both builders."""
    from pathtracer_amd import host
    meshes, mats = _awkward_scene()
    w = O.World.from_meshes(meshes, mats)
    hs = host.Scene.from_meshes(meshes, mats)
    gv = meshes[0][0]
    o_v = np.zeros_like(gv)
    rng = np.random.default_rng(5)
    mid = (0.5 * (gv[rng.integers(0, len(gv), 4000)] + gv[rng.integers(0, len(gv), 4000)])).astype(np.float32)
    sets = {
        "vertices": (o_v, gv.copy()),
        "vertices_off_axis": (np.tile(np.float32([0.25, -0.5, 0.125]), (len(gv), 1)), gv - np.float32([0.25, -0.5, 0.125])),
        "segments": (np.zeros_like(mid), mid),
        "axes": (np.tile(np.float32([0, 0, 0]), (6, 1)), np.float32([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [1, 0, 1], [0, 1, 1]])),
        "primary": util.primary_rays(O.default_camera(), 96, 96),
        "random": util.random_rays(20000, 9, (-2, -2, -1), (2, 2, 5)),
        "axis": util.axis_rays(4000, 10, (-2, -2, -1), (2, 2, 5)),
    }
    for ds_label, ds in (("oracle-built", api.DeviceScene(w.export())), ("host-built", hs.device_scene(0))):
        for label, (o, d) in sets.items():
            o = np.ascontiguousarray(o, np.float32)
            d = np.ascontiguousarray(d, np.float32)
            ref = w.castray(o, d)
            for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
                check_equal(ref, ds.trace_closest(o, d, mode), "awkward/%s/%s/mode%d" % (ds_label, label, mode))
                assert ((ref["tri"] >= 0) == ds.trace_any(o, d, mode)).all(), "awkward/%s/%s any-hit mode %d" % (ds_label, label, mode)
        if ds_label == "oracle-built":
            ds.close()
    assert (w.castray(*[np.ascontiguousarray(a, np.float32) for a in sets["vertices"]])["tri"] >= 0).mean() > 0.9
    hs.close()


def test_awkward_geometry_renders_like_the_oracle(api):
    """The same scene rendered (its odd mesh is the light: zero-area, huge and tiny triangles in the area
    table of Polygon::sample, object.cpp:513-523): radiance sums equal the oracle's bit for bit wherever they
    are finite, and are non-finite in the same pixels."""
    meshes, mats = _awkward_scene()
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    W, H, spp = 96, 96, 4
    ref, rc = w.render(O.default_camera(), W, H, spp, seed=3, rng=0, form=1)
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_EXACT):
        got, gc = ds.render(api.make_camera(), W, H, spp, seed=3, opt=api.make_options(trace_mode=mode))
        fin_r, fin_g = np.isfinite(ref).all(-1), np.isfinite(got).all(-1)
        assert (fin_r == fin_g).all()
        same = (util.bits(ref) == util.bits(got)).all(-1) | ~fin_r
        print("awkward scene, mode %d: %d of %d pixels differ from the oracle" % (mode, (~same).sum(), same.size))
        assert same.all(), "%d of %d pixels differ" % ((~same).sum(), same.size)
        assert int(gc.rays_closest) == int(rc.rays_closest)
        assert gc.paths == rc.paths
    ds.close()


def _soup(seed):
    rng = np.random.default_rng(1000 + seed)
    n_mesh = int(rng.integers(1, 9))
    scale = float(10.0 ** rng.integers(-2, 3))
    meshes, mats = [], []
    for m in range(n_mesh):
        nt = int(rng.integers(1, 400))
        c = rng.uniform(-1, 1, (nt, 1, 3))
        v = (c + rng.normal(0, rng.uniform(0.02, 0.5), (nt, 3, 3))) * scale
        if rng.random() < 0.5:
            v = np.round(v / (0.125 * scale)) * (0.125 * scale)       # coincident vertices, edges, triangles
        k = rng.random(nt) < 0.05
        v[k, 2] = v[k, 1]                                             # zero area
        if m > 0 and rng.random() < 0.5:
            prev = meshes[int(rng.integers(0, m))][0].reshape(-1, 3, 3)
            v[: min(nt, len(prev))] = prev[: min(nt, len(prev))]     # the same triangles in two meshes
        v = np.ascontiguousarray(v.reshape(-1, 3), np.float32)
        meshes.append((v, np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3), m))
        mats.append(((0.5, 0.5, 0.5), (2.0, 2.0, 2.0) if m == 0 else (0, 0, 0)))   # the first mesh glows
    return rng, scale, meshes, mats


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_triangle_soups_render_like_the_oracle(api, seed):
    """The same soups rendered from inside (mesh 0 is the light, with its share of zero-area triangles in
    the area table): radiance sums equal the oracle's bit for bit wherever they are finite, non-finite in the
    same pixels, in fast and in exact traversal."""
    rng, scale, meshes, mats = _soup(seed)
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    W, H, spp = 64, 64, 4
    cam = dict(origin=(0.1 * scale, -0.05 * scale, -1.5 * scale))
    ref, rc = w.render(O.default_camera(**cam), W, H, spp, seed=seed, rng=0, form=1)
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_EXACT):
        got, gc = ds.render(api.make_camera(**cam), W, H, spp, seed=seed, opt=api.make_options(trace_mode=mode))
        fin_r, fin_g = np.isfinite(ref).all(-1), np.isfinite(got).all(-1)
        assert (fin_r == fin_g).all()
        same = (util.bits(ref) == util.bits(got)).all(-1) | ~fin_r
        print("soup %d, mode %d: %d of %d pixels differ from the oracle" % (seed, mode, (~same).sum(), same.size))
        assert same.all(), "soup %d mode %d: %d of %d pixels differ" % (seed, mode, (~same).sum(), same.size)
        assert int(gc.rays_closest) == int(rc.rays_closest)
    ds.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 79, 130])
def test_random_triangle_soups_bit_exact(api, seed):
    """Seeded triangle soups of random size, scale and mesh count (some triangles repeated in another mesh,
    some collapsed to a line, some with vertices snapped to a coarse grid so that edges and whole triangles
    coincide): closest hit and any hit of every traversal mode equal the oracle's on random rays, on rays
    aimed at vertices, on rays along the axes and on two generations of rays that start on the surfaces with
    direction vectors of any length.  Seeds 79 and 130 are the two soups in which a wider search
    (tools/dbg/fuzz_soup.py, 300 seeds) found the fast mode wrong on one such ray each: a hit at t = 4e-6 in a
    leaf box that a ray with |d| = 4.8 had left before the box interval's start, and a hit skipped by distance
    because its leaf box had been clipped, without margin, to a face of the gate that the computed hit point
    lay 1e-5 outside of."""
    rng, scale, meshes, mats = _soup(seed)
    w = O.World.from_meshes(meshes, mats)
    ds = api.DeviceScene(w.export())
    allv = np.concatenate([m[0] for m in meshes])
    n = 20000
    o = (rng.uniform(-2, 2, (n, 3)) * scale).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    tgt = allv[rng.integers(0, len(allv), n)]
    sets = {"random": (o, d), "at_vertices": (o, (tgt - o).astype(np.float32))}
    so, sd = sets["at_vertices"]
    for gen in range(2):  # the random draws above and here are those of tools/dbg/fuzz_soup.py
        ref = w.castray(np.ascontiguousarray(so), np.ascontiguousarray(sd))
        hit = ref["tri"] >= 0
        if hit.sum() < 10:
            break
        p = (so[hit] + sd[hit] * ref["t"][hit, None]).astype(np.float32)   # x = o + d*t in float, camera.cpp:68
        nd = rng.standard_normal(p.shape).astype(np.float32)
        sets["surface%d" % gen] = (p, nd)
        so, sd = p, nd
    sets["axes"] = util.axis_rays(3000, seed, tuple([-2 * scale] * 3), tuple([2 * scale] * 3))
    for label, (ro, rd) in sets.items():
        ro = np.ascontiguousarray(ro, np.float32)
        rd = np.ascontiguousarray(rd, np.float32)
        ref = w.castray(ro, rd)
        for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
            check_equal(ref, ds.trace_closest(ro, rd, mode), "soup%d/%s/mode%d" % (seed, label, mode))
            assert ((ref["tri"] >= 0) == ds.trace_any(ro, rd, mode)).all(), "soup%d/%s any-hit mode %d" % (seed, label, mode)
    ds.close()


@pytest.mark.parametrize("name", ["box40", "atrium_small", "blob_small"])
def test_origins_far_from_the_scene_match_in_every_mode(api, name):
    """The pruning walks carry margins for origins near the scene only (the rounding error of a box test grows
    with the origin's distance).  The ray queries therefore walk every ray whose origin lies beyond 8 times the
    scene's largest coordinate exactly, ray by ray, whatever mode was asked for: rays aimed at the scene from
    1 to 100 000 scene extents away, mixed in one call, equal Scene::castray in all three modes (closest hit,
    any hit and HitData::hits), through the host-pointer and the device-pointer entry points."""
    import torch
    w, ds = util.world(name), dev_scene(api, name)
    verts = w.export()["tri_verts"].reshape(-1, 3)
    rng = np.random.default_rng(4)
    n = 20000
    os_, ds_ = [], []
    for far in (1, 10, 100, 1000, 10000, 100000):
        o = (rng.standard_normal((n, 3)) * far * 3 + np.float32([0, 0, 2])).astype(np.float32)
        tgt = verts[rng.integers(0, len(verts), n)] + rng.normal(0, 0.05, (n, 3))
        os_.append(o)
        ds_.append((tgt - o).astype(np.float32))
    o, d = np.concatenate(os_), np.concatenate(ds_)
    perm = rng.permutation(len(o))          # near and far origins side by side in every wave
    o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
    ref = w.castray(o, d)
    assert (ref["tri"] >= 0).mean() > 0.3
    ref_hits = w.castray(o, d, want_hits=True)["hits"]
    for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
        check_equal(ref, ds.trace_closest(o, d, mode), "%s/far origins/mode %d" % (name, mode))
        assert ((ref["tri"] >= 0) == ds.trace_any(o, d, mode)).all()
        assert (ds.trace_hits(o, d, mode) == ref_hits).all()
    # device pointers: both launches are enqueued without looking at the origins
    dev = torch.device("cuda", 0)
    cols = [torch.from_numpy(np.ascontiguousarray(a[:, i])).to(dev) for a in (o, d) for i in range(3)]
    t = torch.empty(len(o), dtype=torch.float32, device=dev)
    tri = torch.empty(len(o), dtype=torch.int32, device=dev)
    u = torch.empty_like(t)
    v = torch.empty_like(t)
    ds.trace_closest_dev(len(o), [c.data_ptr() for c in cols], [t.data_ptr(), tri.data_ptr(), u.data_ptr(), v.data_ptr()],
                         mode=api.PT_TRACE_FAST)
    torch.cuda.synchronize()
    check_equal(ref, dict(t=t.cpu().numpy(), tri=tri.cpu().numpy(), u=u.cpu().numpy(), v=v.cpu().numpy()),
                "%s/far origins/device pointers" % name)


def test_description_with_boxes_that_are_not_nested_is_walked_exactly(api):
    """PT_TRACE_FAST tests the lowest internal box above a triangle in place of all of them; that is the
    reference's test only for nested boxes.  A description with an inner box refitted too tight (the exact walk
    then loses the triangles that stick out, as the reference would with that tree) is flagged at upload and
    walked exactly in every mode: same hits as PT_TRACE_EXACT, and they differ from the intact scene's."""
    good = util.world("atrium_small").export()
    nodes, off = good["blas_nodes"], good["blas_node_offset"]
    e = {k: v.copy() for k, v in good.items()}
    changed = 0
    for m in range(len(off) - 1):
        if off[m + 1] - off[m] >= 5 and nodes[off[m]]["count"] == 0:
            lo, hi = nodes["bmin"][off[m]].copy(), nodes["bmax"][off[m]].copy()
            e["blas_nodes"]["bmax"][off[m]] = lo + (hi - lo) * 0.5
            changed += 1
    assert changed
    ds_bad, ds_good = api.DeviceScene(e), dev_scene(api, "atrium_small")
    assert ds_bad.info["boxes_nested"] == 0 and ds_good.info["boxes_nested"] == 1
    o, d = util.random_rays(60000, 8)
    exact = ds_bad.trace_closest(o, d, api.PT_TRACE_EXACT)
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
        check_equal(exact, ds_bad.trace_closest(o, d, mode), "not nested/mode %d" % mode)
        assert ((exact["tri"] >= 0) == ds_bad.trace_any(o, d, mode)).all()
    intact = ds_good.trace_closest(o, d, api.PT_TRACE_EXACT)
    assert (intact["tri"] != exact["tri"]).mean() > 0.01      # the shrunk boxes do change what the reference's walk finds
    cam = api.make_camera(vw=16, vh=9)
    a, _ = ds_bad.render(cam, 96, 54, 3, seed=2, opt=api.make_options(trace_mode=api.PT_TRACE_EXACT))
    b, _ = ds_bad.render(cam, 96, 54, 3, seed=2, opt=api.make_options(trace_mode=api.PT_TRACE_FAST))
    assert (util.bits(a) == util.bits(b)).all()
    ds_bad.close()


def test_hit_with_a_nan_distance_is_never_recorded(api):
    """Known deviation, pinned: Triangle::intersect can return a hit whose t is NaN (a determinant so small that
    its reciprocal overflows, times an exactly zero numerator: inf * 0).  In the reference what becomes of it
    depends on where the triangle sits in its leaf and tree (every comparison with NaN is false); here such a
    hit is never recorded, in any mode: the ray's answer is that of the scene without that triangle."""
    from pathtracer_amd import scenes
    meshes, mats = scenes.box_room(10)
    tiny = np.float32([[0, 0, 0], [1e-20, 0, 0], [0, 1e-20, 0]])
    meshes2 = list(meshes) + [(tiny, np.uint32([[0, 1, 2]]), 0)]
    w0, w1 = O.World.from_meshes(meshes, mats), O.World.from_meshes(meshes2, mats)
    o = np.float32([[0, 0, -0.5], [0, 0, 0.5], [0.25, 0, -0.5]])
    d = np.float32([[0, 0, 1], [0, 0, -1], [0, 0, 1]])
    # the oracle's Triangle::intersect does report the NaN hit for the first two rays
    one = O.World.from_meshes([(tiny, np.uint32([[0, 1, 2]]), 0)], mats)
    r = one.castray(o, d)
    assert np.isnan(r["t"][:2]).all() and (r["tri"][:2] == 0).all() and r["tri"][2] == -1
    ds0, ds1 = api.DeviceScene(w0.export()), api.DeviceScene(w1.export())
    for mode in (api.PT_TRACE_EXACT, api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
        a, b = ds0.trace_closest(o, d, mode), ds1.trace_closest(o, d, mode)
        assert (util.bits(a["t"]) == util.bits(b["t"])).all() and (a["tri"] == b["tri"]).all()
        assert not np.isnan(b["t"]).any() and b["tri"][0] >= 0      # the second ray leaves through the room's open side
    ds0.close()
    ds1.close()


def test_garbage_hit_of_a_ray_in_a_triangles_plane_is_a_known_deviation(api):
    """Known deviation of the pruning modes, pinned.  A ray that runs in the plane of a triangle to within rounding
    (here: aimed at one of its vertices from far away, determinant -5e-9 where its rounding error is as large)
    makes Triangle::intersect (object.cpp:406-429) compute u, v and t from noise; about one such ray in 10^7
    lands inside the tests by accident and the reference reports a hit at a distance that has nothing to do with
    where the triangle is (here t = 0.957 for a triangle the ray touches at t = 1.000, 1.2 units farther on; in
    double precision the ray misses it: u = 1.9, v = 49).  The exact mode tests every triangle of every leaf the
    ray passes, as the reference does, and returns that hit.  The pruning modes skip what begins beyond the best
    hit they hold, and when the true closest hit (t = 0.9713) is found first the triangle's leaf, which begins at
    t = 1, is never opened.  tools/dbg/near_limit.py measures the rate: 3 rays of 68 million aimed at vertices,
    none among the rays of any other probe or test."""
    w, ds = util.world("blob_small"), dev_scene(api, "blob_small")
    o = np.array([[0xc1dd8081, 0xc0a3536a, 0x408aafef]], np.uint32).view(np.float32)
    d = np.array([[0x41dddba3, 0x40d21d38, 0xbf96037e]], np.uint32).view(np.float32)
    ref = w.castray(o, d)
    assert ref["tri"][0] == 1395 and ref["t"][0] == np.float32(0.95703125) and ref["v"][0] == 0.0
    exact = ds.trace_closest(o, d, api.PT_TRACE_EXACT)
    assert exact["tri"][0] == 1395 and util.bits(exact["t"])[0] == util.bits(ref["t"])[0]
    for mode in (api.PT_TRACE_FAST, api.PT_TRACE_FAST_BINARY):
        got = ds.trace_closest(o, d, mode)
        # the hit that is really there (with the default tree and settings), or the reference's when the walk
        # happens to open that leaf first (other leaf sizes or child orders)
        real = got["tri"][0] == 3275 and abs(float(got["t"][0]) - 0.9713243) < 1e-6
        noise = got["tri"][0] == 1395 and util.bits(got["t"])[0] == util.bits(ref["t"])[0]
        assert real or noise

