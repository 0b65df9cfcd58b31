"""GPU render parity with the oracle.  Both sides use the same counter-based random
numbers and the same operation order, so per-pixel radiance sums are compared as bit
patterns; the few paths that diverge (a double-precision libm result that rounds to a
different float) are bounded as a fraction."""
import numpy as np
import pytest

from tests import util
from oracle import pt_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from pathtracer_amd import api
    assert api.device_count() >= 1
    return api


def both(api, name, W, H, spp, cam_kw=None, opt_kw=None, seed=5):
    cam_kw = cam_kw or {}
    opt_kw = opt_kw or {}
    w = util.world(name)
    ds = api.DeviceScene(w.export())
    ref, rc = w.render(O.default_camera(**cam_kw), W, H, spp, seed=seed, opt=O.default_options(**opt_kw), rng=0, form=1)
    got, gc = ds.render(api.make_camera(**cam_kw), W, H, spp, seed=seed, opt=api.make_options(**opt_kw))
    ds.close()
    return ref, rc, got, gc


def assert_images_match(ref, got, max_diff_frac=0.0):
    assert np.isfinite(got).all() == np.isfinite(ref).all()
    same = (util.bits(ref) == util.bits(got)).all(-1)
    frac = 1.0 - same.mean()
    print("pixels that differ from the oracle: %d of %d (allowed %g)" % ((~same).sum(), same.size, max_diff_frac))  # shown with -s / -rP
    assert frac <= max_diff_frac, "%.4f%% of pixels differ" % (100 * frac)
    m_ref, m_got = float(ref.mean()), float(got.mean())
    assert abs(m_ref - m_got) <= 0.01 * abs(m_ref) + 1e-6


def test_plane_renders_black(api):
    ref, rc, got, gc = both(api, "plane", 256, 256, 16)
    assert got.sum() == 0 and ref.sum() == 0
    assert gc.paths == rc.paths == 256 * 256 * 16
    assert gc.rays_closest == rc.rays_closest and gc.rays_shadow == rc.rays_shadow == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_box_matches_oracle_bitwise(api, mode):
    ref, rc, got, gc = both(api, "box", 160, 90, 6, cam_kw=dict(vw=16, vh=9), opt_kw=dict(trace_mode=mode))
    assert_images_match(ref, got)
    assert gc.rays_closest == rc.rays_closest
    assert gc.rays_shadow == rc.rays_shadow


def test_many_lights_matches_oracle(api):
    ref, rc, got, gc = both(api, "terrain_small", 96, 54, 3, cam_kw=dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9))
    assert_images_match(ref, got)


@pytest.mark.parametrize("lights", [64, 96, 128])
def test_whole_vertex_visibility_walks_match_oracle(api, monkeypatch, lights):
    """Split shading with as many lights as a wave has lanes (or twice as many): a wave walks one vertex's
    visibility rays at a time.  96 lights do not fill whole waves and take the ordinary path.  Bitwise against
    the oracle, ray counts included."""
    from pathtracer_amd import scenes
    monkeypatch.setenv("PT_NO_RESIDENT", "1")          # a small scene through the kernels of the large ones
    w = O.World.from_meshes(*scenes.terrain_lights(24, lights))
    ds = api.DeviceScene(w.export())
    assert ds.info["n_lights"] == lights
    kw = dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9)
    ref, rc = w.render(O.default_camera(**kw), 96, 54, 2, seed=9, rng=0, form=1)
    got, gc = ds.render(api.make_camera(**kw), 96, 54, 2, seed=9)
    ds.close()
    assert_images_match(ref, got)
    assert gc.rays_closest == rc.rays_closest and gc.rays_shadow == rc.rays_shadow > 0


def test_extensions_match_oracle(api):
    """max_depth, cosine sampling, thin lens and a background colour: none of them is in the
    reference (parity unpinned); the oracle implements the same rules."""
    ref, rc, got, gc = both(api, "atrium_small", 128, 72, 4, cam_kw=dict(vw=16, vh=9, aperture=0.05, focus=3.0),
                            opt_kw=dict(max_depth=5, sampling=1, bgcolor=(0.1, 0.2, 0.3)))
    assert_images_match(ref, got)
    assert gc.bounce_passes <= 6 * 1


@pytest.mark.parametrize("name,cam_kw,opt_kw", [
    ("box", dict(vw=16, vh=9), dict()),
    ("box", dict(vw=16, vh=9), dict(trace_mode=2)),
    ("atrium_small", dict(vw=16, vh=9, aperture=0.05, focus=3.0), dict(max_depth=5)),
    ("terrain_small", dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9), dict()),
])
def test_tail_kernel_and_wavefront_passes_agree(api, monkeypatch, name, cam_kw, opt_kw):
    """The last paths of a batch are followed to their ends by one kernel instead of one pass per depth
    level.  Whatever the hand-over point — never (PT_TAIL=0), the default, after two levels' worth of
    attrition, or from the first vertex on — the radiance sums are the same bit for bit, the ray counts
    are equal, and they are the oracle's."""
    w = util.world(name)
    W, H, spp = 192, 108, 4
    ref, rc = w.render(O.default_camera(**cam_kw), W, H, spp, seed=13, opt=O.default_options(**{k: v for k, v in opt_kw.items() if k != "trace_mode"}), rng=0, form=1)
    imgs, cnts = [], []
    for tail in ("0", None, "30000", str(1 << 30)):
        if tail is None:
            monkeypatch.delenv("PT_TAIL", raising=False)
        else:
            monkeypatch.setenv("PT_TAIL", tail)   # read when the scene is uploaded
        ds = api.DeviceScene(w.export())
        img, c = ds.render(api.make_camera(**cam_kw), W, H, spp, seed=13, opt=api.make_options(**opt_kw))
        ds.close()
        imgs.append(img)
        cnts.append(c)
    monkeypatch.delenv("PT_TAIL", raising=False)
    for img, c in zip(imgs[1:], cnts[1:]):
        assert (util.bits(img) == util.bits(imgs[0])).all()
        assert (c.paths, c.rays_closest, c.rays_shadow, c.bounce_passes) == (cnts[0].paths, cnts[0].rays_closest, cnts[0].rays_shadow, cnts[0].bounce_passes)
    assert_images_match(ref, imgs[0])
    assert int(cnts[0].rays_closest) == int(rc.rays_closest)


def test_sample_split_is_exact(api):
    """Any split of the sample range over calls gives the same sums as long as the order of
    the additions per pixel is kept: 0..7 in one call == 0..3 then 4..7."""
    w = util.world("box")
    ds = api.DeviceScene(w.export())
    cam = api.make_camera(vw=16, vh=9)
    a, _ = ds.render(cam, 128, 72, 8, seed=3)
    b, _ = ds.render(cam, 128, 72, 4, spp_begin=0, seed=3)
    b, _ = ds.render(cam, 128, 72, 4, spp_begin=4, seed=3, accum=b)
    assert (util.bits(a) == util.bits(b)).all()
    # strided shards (what each of G devices renders) sum to the same image up to float reordering
    s0, _ = ds.render(cam, 128, 72, 4, spp_begin=0, seed=3, opt=api.make_options(spp_stride=2))
    s1, _ = ds.render(cam, 128, 72, 4, spp_begin=1, seed=3, opt=api.make_options(spp_stride=2))
    np.testing.assert_allclose(s0 + s1, a, rtol=2e-5, atol=1e-6)
    # batch size does not change anything
    c, _ = ds.render(cam, 128, 72, 8, seed=3, opt=api.make_options(batch_spp=3))
    assert (util.bits(a) == util.bits(c)).all()
    ds.close()


def test_statistics_against_reference_generator(api):
    """Against the oracle driven by mt19937 in the reference's recursive form: a different random
    stream, so only statistics can agree.  Tolerance: relative difference of the image means
    <= 3 % at 64 spp on 96x54 (two oracle runs with different seeds differ by about 1 %)."""
    w = util.world("box")
    ds = api.DeviceScene(w.export())
    W, H, spp = 96, 54, 64
    ref, _ = w.render(O.default_camera(vw=16, vh=9), W, H, spp, seed=11, rng=1, form=0)
    got, _ = ds.render(api.make_camera(vw=16, vh=9), W, H, spp, seed=12)
    ds.close()
    assert np.isfinite(got).all()
    assert abs(ref.mean() - got.mean()) <= 0.03 * ref.mean()


@pytest.mark.parametrize("W,H,spp", [(96, 54, 64), (192, 108, 32)])
def test_noise_aware_image_parity_with_the_reference_estimator(api, W, H, spp):
    """Converged-image parity against the oracle in its reference configuration (recursive estimator,
    mt19937): different random streams, so the comparison is statistical.  Stated tolerances: (i) no
    non-finite pixel; (ii) image means within 2 %; (iii) per-pixel RMSE(GPU, CPU) within [0.9, 1.1] of
    RMSE(CPU, CPU') of two independent CPU renders; (iv) 16x16 block means within 4 standard deviations of the
    mean of eight independent CPU renders for at least 99 % of blocks, the standard deviation of a block
    estimated from those eight renders (a GPU render is one more draw: variance (1 + 1/8) sigma^2)."""
    w = util.world("box")
    ds = api.DeviceScene(w.export())
    cam_o, cam_g = O.default_camera(vw=16, vh=9), api.make_camera(vw=16, vh=9)
    cpu = [w.render(cam_o, W, H, spp, seed=21 + k, rng=1, form=0)[0] / spp for k in range(8)]
    g, _ = ds.render(cam_g, W, H, spp, seed=23)
    ds.close()
    g = g / spp
    a, b = cpu[0], cpu[1]
    assert np.isfinite(g).all()
    assert abs(g.mean() - a.mean()) <= 0.02 * a.mean()
    rmse = lambda x, y: float(np.sqrt(((x - y) ** 2).mean()))
    ratio = rmse(g, a) / rmse(a, b)
    assert 0.9 <= ratio <= 1.1, ratio
    by, bx = H // 16, W // 16
    def blocks(x):
        return x[:by * 16, :bx * 16].reshape(by, 16, bx, 16, 3).mean((1, 3))
    B = np.stack([blocks(x) for x in cpu])
    mean, sd = B.mean(0), B.std(0, ddof=1) + 1e-3 * B.mean()
    z = np.abs(blocks(g) - mean) / (np.sqrt(1 + 1 / 8) * sd)
    assert (z <= 4).mean() >= 0.99, float((z <= 4).mean())


def test_full_size_frame_properties(api):
    """The benchmark's own configuration (1 002-triangle room, 1920x1080): properties that need no
    CPU reference at that size.  Splitting the sample range over two calls reproduces one call bit for
    bit; strided shards add up; ray counts are the same in every run; nothing is non-finite."""
    from pathtracer_amd import host, scenes
    hs = host.Scene.from_meshes(*scenes.box_room(10))
    ds = hs.device_scene(0)
    cam = api.make_camera(vw=16, vh=9)
    W, H = 1920, 1080
    one, c1 = ds.render(cam, W, H, 4, seed=8)
    two, _ = ds.render(cam, W, H, 2, seed=8)
    two, _ = ds.render(cam, W, H, 2, spp_begin=2, seed=8, accum=two)
    assert (util.bits(one) == util.bits(two)).all()
    s0, c0 = ds.render(cam, W, H, 2, spp_begin=0, seed=8, opt=api.make_options(spp_stride=2))
    s1, c2 = ds.render(cam, W, H, 2, spp_begin=1, seed=8, opt=api.make_options(spp_stride=2))
    np.testing.assert_allclose(s0 + s1, one, rtol=2e-5, atol=1e-6)
    assert c0.rays_closest + c2.rays_closest == c1.rays_closest
    assert c0.rays_shadow + c2.rays_shadow == c1.rays_shadow
    assert np.isfinite(one).all() and c1.paths == W * H * 4
    assert (one.sum(-1) > 0).mean() > 0.5


@pytest.mark.parametrize("mode", [0, 1])
def test_lds_resident_and_split_pipelines_agree(api, mode, monkeypatch):
    """A scene that fits LDS is rendered by the fused shading kernel with chunk-wise walks; a scene
    that does not fit goes through the split pipeline (visibility rays prepared, walked with
    per-lane refill, vertices finished).  Forcing the second on a small scene must give the same
    radiance sums bit for bit, and the same ray counts."""
    w = util.world("box")
    cam = api.make_camera(vw=16, vh=9)
    opt = api.make_options(trace_mode=mode)
    ds = api.DeviceScene(w.export())
    a, ca = ds.render(cam, 160, 90, 5, seed=9, opt=opt)
    ds.close()
    monkeypatch.setenv("PT_NO_RESIDENT", "1")  # read when the scene is uploaded
    ds = api.DeviceScene(w.export())
    b, cb = ds.render(cam, 160, 90, 5, seed=9, opt=opt)
    o, d = util.random_rays(20000, 3)
    hit_b = ds.trace_closest(o, d, mode=mode)
    any_b = ds.trace_any(o, d, mode=mode)
    ds.close()
    monkeypatch.delenv("PT_NO_RESIDENT")
    ds = api.DeviceScene(w.export())
    hit_a = ds.trace_closest(o, d, mode=mode)
    any_a = ds.trace_any(o, d, mode=mode)
    ds.close()
    assert (util.bits(a) == util.bits(b)).all()
    assert ca.rays_closest == cb.rays_closest and ca.rays_shadow == cb.rays_shadow and ca.paths == cb.paths
    for key in ("t", "u", "v"):
        assert (util.bits(hit_a[key]) == util.bits(hit_b[key])).all()
    assert (hit_a["tri"] == hit_b["tri"]).all()
    assert (np.asarray(any_a) == np.asarray(any_b)).all()


def test_cached_scene_renders_identically(api, tmp_path):
    """A scene loaded from its binary cache (no parse, no tree build) gives the same image, bit for
    bit, and the same closest hits as the scene it was saved from."""
    from pathtracer_amd import host, scenes
    meshes, mats = scenes.atrium(24, 6, 16, 8)
    s = host.Scene.from_meshes(meshes, mats)
    p = str(tmp_path / "atrium.ptscene")
    s.save_cache(p)
    c = host.Scene.from_cache(p)
    cam = api.make_camera(vw=16, vh=9)
    a, ca = s.device_scene(0).render(cam, 128, 72, 4, seed=3)
    b, cb = c.device_scene(0).render(cam, 128, 72, 4, seed=3)
    assert (util.bits(a) == util.bits(b)).all()
    assert ca.rays_closest == cb.rays_closest and ca.rays_shadow == cb.rays_shadow
    o, d = util.random_rays(5000, 11)
    ha, hb = s.device_scene(0).trace_closest(o, d), c.device_scene(0).trace_closest(o, d)
    assert (ha["tri"] == hb["tri"]).all() and (util.bits(ha["t"]) == util.bits(hb["t"])).all()


@pytest.mark.parametrize("sampling", [0, 1])
@pytest.mark.parametrize("split", [False, True])
def test_textured_scene_matches_oracle_bitwise(api, tmp_path, monkeypatch, sampling, split):
    """Textured diffuse (camera.cpp:121-146): barycentric texture coordinates, nearest texel, the
    texel itself as brdf.  Same radiance bits as the oracle, through the fused shading kernel and
    through the split pipeline, with the reference's sampling and with the cosine extension."""
    from pathtracer_amd import scenes
    meshes, mats = scenes.box_room(6)
    rng = np.random.default_rng(2)
    checker = np.zeros((8, 8, 3), np.uint8)
    checker[::2, ::2] = (230, 40, 40)
    checker[1::2, 1::2] = (40, 40, 230)
    checker[::2, 1::2] = (220, 220, 220)
    checker[1::2, ::2] = (30, 160, 30)
    tex = {0: checker, 2: rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)}
    p = str(tmp_path / "room.obj")
    scenes.write_textured_obj(p, meshes, mats, tex)
    w = O.World.from_obj(p)
    if split:
        monkeypatch.setenv("PT_NO_RESIDENT", "1")
    ds = api.DeviceScene(w.export())
    cam_kw = dict(vw=16, vh=9)
    ref, rc = w.render(O.default_camera(**cam_kw), 160, 90, 6, seed=4, opt=O.default_options(sampling=sampling), rng=0, form=1)
    got, gc = ds.render(api.make_camera(**cam_kw), 160, 90, 6, seed=4, opt=api.make_options(sampling=sampling))
    ds.close()
    assert_images_match(ref, got)
    assert gc.rays_closest == rc.rays_closest
    # the maps matter: the same room without them renders differently
    plain = O.World.from_meshes(meshes, mats)
    ref0, _ = plain.render(O.default_camera(**cam_kw), 160, 90, 6, seed=4, opt=O.default_options(sampling=sampling), rng=0, form=1)
    assert np.abs(ref0 - ref).mean() > 0.01 * np.abs(ref0).mean()


@pytest.mark.parametrize("name,cam_kw", [
    ("box", dict(vw=16, vh=9)),
    ("atrium_small", dict(vw=16, vh=9)),
    ("terrain_small", dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9)),
])
def test_larger_frame_matches_oracle_bitwise(api, name, cam_kw):
    """640x360 at 4 spp (0.9 M paths, tens of millions of rays) on a scene in LDS, one outside it and
    one with 20 lights: no pixel may differ in any bit, and the ray counts must be equal."""
    ref, rc, got, gc = both(api, name, 640, 360, 4, cam_kw=cam_kw, seed=7)
    assert_images_match(ref, got)
    assert gc.rays_closest == rc.rays_closest and gc.rays_shadow == rc.rays_shadow


def test_camera_far_from_the_scene_still_matches_the_oracle(api):
    """A box test's rounding error grows with the distance of the ray's origin; the pruning walks carry a
    margin for origins near the scene only, so a render whose camera stands farther away than 8 times the
    scene's largest coordinate runs the exact walk by itself.  Camera 5 000 units in front of the room's open
    side with a field of view that just frames it: radiance sums equal the oracle's."""
    kw = dict(origin=(0.0, 0.0, -5000.0), fov=0.06, vw=16, vh=9)
    ref, rc, got, gc = both(api, "box", 96, 54, 4, cam_kw=kw, opt_kw=dict(trace_mode=api.PT_TRACE_FAST))
    assert_images_match(ref, got)
    assert rc.rays_shadow > 0 and gc.rays_closest == rc.rays_closest
