"""GPU tests of the boundary itself: error behaviour of the C ABI, the host-side Scene/Camera mirror
driving it, the device-pointer entry points and the headless driver."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import util
from oracle import pt_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def api():
    from pathtracer_amd import api
    assert api.device_count() >= 1
    return api


def test_malformed_descriptions_are_rejected(api):
    good = util.world("box").export()

    def broken(**changes):
        e = {k: v.copy() for k, v in good.items()}
        for k, f in changes.items():
            f(e[k])
        return e

    def set_(idx, field, val):
        def f(a):
            a[field][idx] = val
        return f

    def assign(idx, val):
        def f(a):
            a[idx] = val
        return f

    cases = [
        broken(blas_nodes=set_(0, "index_or_first", 10 ** 6)),     # child index out of range
        broken(blas_order=assign(0, 10 ** 6)),                      # triangle id out of range
        broken(blas_order=assign(1, int(good["blas_order"][0]))),   # a triangle twice, another never
        broken(tlas_order=assign(0, 77)),                           # mesh id out of range
        broken(mesh_material=assign(0, 99)),                        # material out of range
        broken(blas_nodes=set_(1, "index_or_first", 1)),            # node referenced twice / cycle
    ]
    for e in cases:
        with pytest.raises(api.PtError) as ei:
            api.DeviceScene(e)
        assert ei.value.code == -1 and str(ei.value)
    with pytest.raises(api.PtError):
        api.DeviceScene(good, device=99)


def test_null_and_bad_arguments(api):
    L = api.lib()
    ds = api.DeviceScene(util.world("box").export())
    assert L.pt_scene_info(None, None) == -1
    assert L.pt_trace_closest(ds.handle, 4, None, None, None, None, None, None, None, None, None, None, 0) == -1
    assert b"null" in L.pt_last_error()
    o, d = util.random_rays(16, 0)
    with pytest.raises(api.PtError):
        ds.trace_closest(o, d, mode=7)
    cam = api.make_camera()
    with pytest.raises(api.PtError):
        ds.render(cam, 0, 10, 1, accum=np.zeros((10, 0, 3), np.float32))
    with pytest.raises(api.PtError):
        ds.render(api.make_camera(aperture=0.1, focus=0.0), 8, 8, 1)
    with pytest.raises(api.PtError):
        ds.render(cam, 8, 8, 1, opt=api.make_options(rr_prob=0.0))
    r = ds.trace_closest(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(r["t"]) == 0
    acc, cnt = ds.render(cam, 8, 8, 0)
    assert acc.sum() == 0 and cnt.paths == 0
    ds.close()


def test_host_scene_renders_like_the_abi(api):
    from pathtracer_amd import host, scenes
    meshes, mats = scenes.box_room(10)
    hs = host.Scene.from_meshes(meshes, mats)
    ds = hs.device_scene(0)
    cam = api.make_camera(vw=16, vh=9)
    W, H, spp = 80, 45, 5
    calls = []
    pixels, cnt = hs.render(cam, W, H, spp, seed=3, callback=lambda: calls.append(1), every=1)
    assert len(calls) == spp                        # cb() after every pass, camera.cpp:196-197
    sums, c2 = ds.render(cam, W, H, spp, seed=3)
    # the running mean (i*p + 255*L)/(i+1) and sum/n*255 round differently: equal to float rounding, every pixel
    assert np.isclose(pixels, sums / spp * 255, rtol=1e-5, atol=1e-3).all()
    assert cnt.paths == c2.paths and cnt.rays_closest == c2.rays_closest
    # and the oracle's Camera::pixels (camera.cpp:188-190) bit for bit: the accumulator is the reference's own
    ref, _ = O.World.from_meshes(meshes, mats).render(O.default_camera(vw=16, vh=9), W, H, spp, seed=3,
                                                       running_mean=True, form=1)
    assert (util.bits(pixels) == util.bits(ref)).all()


def _display_rule(pixels):
    """main.cpp:43: std::clamp((int)double(value), 0, 255) as the x86-64 build converts (toward zero; NaN and
    values outside int -> INT_MIN -> 0)."""
    p = np.asarray(pixels, np.float64)
    bad = ~np.isfinite(p) | (p >= 2147483648.0) | (p < -2147483648.0)
    t = np.where(bad, -2147483648.0, np.trunc(np.where(bad, 0.0, p)))
    return np.clip(t, 0, 255).astype(np.uint8)


def test_progressive_display_is_converted_on_the_device(api):
    """Camera::render with the GUI callback's display conversion (main.cpp:40-52) on the device: after every
    pass the callback sees clamp((int)pixels, 0, 255) of the running image, and the bytes equal that rule
    applied to the oracle's Camera::pixels (bright pixels near the light exceed 255 and clamp; the rule is
    also checked on values no render produces: negative, NaN, infinite, beyond int)."""
    import torch
    from pathtracer_amd import host, scenes
    meshes, mats = scenes.box_room(10)
    hs = host.Scene.from_meshes(meshes, mats)
    cam = api.make_camera(vw=16, vh=9)
    W, H, spp = 96, 54, 4
    seen = []
    rgb, pixels = hs.render_display(cam, W, H, spp, seed=3, callback=seen.append, every=1)
    assert len(seen) == spp and (seen[-1] == rgb).all()
    w = O.World.from_meshes(meshes, mats)
    for k in range(1, spp + 1):          # the image after k passes = the oracle's Camera::pixels of a k-pass render
        ref, _ = w.render(O.default_camera(vw=16, vh=9), W, H, k, seed=3, rng=0, form=1, running_mean=True)
        assert (seen[k - 1] == _display_rule(ref)).all(), k
    assert (util.bits(pixels) == util.bits(ref)).all()     # Camera::pixels itself, bit for bit
    assert rgb.max() == 255 and rgb.min() == 0
    # and through the plain callback path: `pixels` after every pass
    got = []
    px, _ = hs.render(cam, W, H, spp, seed=3, callback=lambda: got.append(1), every=2)
    assert len(got) == 2 and (util.bits(px) == util.bits(ref)).all()
    # the conversion kernels on hand-made sums
    vals = np.float32([0, 0.5, 1.0, 1.004, 300.0, -3.0, np.nan, np.inf, -np.inf, 3e9, 254.999 / 255, 2.0 / 255] * 2)
    dev = torch.device("cuda", 0)
    src = torch.from_numpy(vals).to(dev)
    out8 = torch.zeros(len(vals), dtype=torch.uint8, device=dev)
    outf = torch.zeros(len(vals), dtype=torch.float32, device=dev)
    api.display_dev(src.data_ptr(), len(vals) // 3, out8.data_ptr())
    torch.cuda.synchronize()
    assert (out8.cpu().numpy() == _display_rule(vals)).all()
    api.resolve_display_dev(src.data_ptr(), len(vals) // 3, 1, out8.data_ptr())
    api.resolve_dev(src.data_ptr(), len(vals), 2, outf.data_ptr())
    torch.cuda.synchronize()
    assert (out8.cpu().numpy() == _display_rule(vals * np.float32(255))).all()
    assert (util.bits(outf.cpu().numpy()) == util.bits(vals / np.float32(2) * np.float32(255))).all()
    assert (util.bits(api.resolve(vals, 2)) == util.bits(outf.cpu().numpy())).all()


def test_headless_driver_display_output(tmp_path):
    """ptrender --display writes what the reference's window would show (main.cpp:43-47), converted on the device."""
    from pathtracer_amd import scenes
    obj = str(tmp_path / "room.obj")
    scenes.write_obj(obj, *scenes.box_room(6))
    exe = os.path.join(ROOT, "pathtracer_amd", "lib", "ptrender")
    out_d, out_p = str(tmp_path / "d.ppm"), str(tmp_path / "p.ppm")
    common = [exe, obj, "--spp", "3", "--width", "64", "--height", "36", "--vw", "16", "--vh", "9", "--seed", "5"]
    r = subprocess.run(common + ["-o", out_d, "--display", "--progress"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "pass 3/3" in r.stdout, r.stderr
    r = subprocess.run(common + ["-o", out_p], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = open(out_d, "rb").read()
    head = b"P6\n64 36\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + 64 * 36 * 3
    disp = np.frombuffer(raw[len(head):], np.uint8)
    plain = np.array(open(out_p).read().split()[4:], np.int64)      # Camera::save: (int)value, no clamp
    assert (disp == np.clip(plain, 0, 255)).all() and plain.max() > 255


def test_device_pointer_entry_points(api):
    import torch
    ds = api.DeviceScene(util.world("atrium_small").export())
    o, d = util.random_rays(5000, 4)
    dev = torch.device("cuda", 0)
    cols = [torch.from_numpy(np.ascontiguousarray(a[:, i])).to(dev) for a in (o, d) for i in range(3)]
    t = torch.empty(len(o), dtype=torch.float32, device=dev)
    tri = torch.empty(len(o), dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ds.trace_closest_dev(len(o), [c.data_ptr() for c in cols], [t.data_ptr(), tri.data_ptr(), 0, 0],
                             mode=api.PT_TRACE_EXACT, stream=s.cuda_stream)
    s.synchronize()
    ref = ds.trace_closest(o, d, api.PT_TRACE_EXACT)
    assert (t.cpu().numpy().view(np.uint32) == util.bits(ref["t"])).all()
    assert (tri.cpu().numpy() == ref["tri"]).all()
    cam = api.make_camera(vw=16, vh=9)
    acc = torch.zeros(40 * 24 * 3, dtype=torch.float32, device=dev)
    ds.render_dev(cam, 40, 24, 3, acc.data_ptr(), seed=2, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host_acc, _ = ds.render(cam, 40, 24, 3, seed=2)
    assert (acc.cpu().numpy().view(np.uint32) == util.bits(host_acc.reshape(-1))).all()
    ds.close()


def test_render_multi_on_one_device_equals_render(api):
    """pt_render_multi with a single device is pt_render bit for bit (same samples, same additions), also as a
    shard of a larger frame and through the host-side Camera."""
    ds = api.DeviceScene(util.world("box").export())
    cam = api.make_camera(vw=16, vh=9)
    a, ca = ds.render(cam, 96, 54, 5, spp_begin=3, seed=4, opt=api.make_options(spp_stride=2))
    b, cb = api.render_multi([ds], cam, 96, 54, 5, spp_begin=3, seed=4, opt=api.make_options(spp_stride=2))
    assert (util.bits(a) == util.bits(b)).all() and ca.as_dict() == cb.as_dict()
    with pytest.raises(api.PtError):
        api.render_multi([ds, ds], cam, 96, 54, 5)      # two copies on one device
    ds.close()


def test_render_multi_rehearsed_on_one_device(api, monkeypatch):
    """The n > 1 branch of pt_render_multi on a box with one GPU: with PT_MULTI_SHARE_DEVICE=1 three copies of the
    scene on device 0 stand in for three devices.  One host thread per copy, the samples dealt round-robin,
    counters merged; only the all-reduce is replaced (RCCL takes a device once per communicator) by an on-device
    sum.  The frame is the one-device frame up to the order of the float additions, ray counts add up exactly,
    and without the switch two scenes on one device are still refused."""
    e = util.world("atrium_small").export()
    copies = [api.DeviceScene(e) for _ in range(3)]
    cam = api.make_camera(vw=16, vh=9)
    one, c1 = copies[0].render(cam, 160, 90, 10, seed=6)
    with pytest.raises(api.PtError):
        api.render_multi(copies, cam, 160, 90, 10, seed=6)
    monkeypatch.setenv("PT_MULTI_SHARE_DEVICE", "1")
    many, cn = api.render_multi(copies, cam, 160, 90, 10, seed=6)
    np.testing.assert_allclose(many, one, rtol=2e-5, atol=1e-6)
    assert (cn.paths, cn.rays_closest, cn.rays_shadow) == (c1.paths, c1.rays_closest, c1.rays_shadow)
    # fewer samples than copies: the idle copies add nothing
    a, _ = copies[0].render(cam, 160, 90, 2, seed=6)
    b, cb = api.render_multi(copies, cam, 160, 90, 2, seed=6)
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-6)
    with pytest.raises(api.PtError):
        api.render_multi([copies[0], copies[0]], cam, 160, 90, 2)     # the same object twice
    for s in copies:
        s.close()


def test_render_multi_all_reduce_runs_through_rccl(api, monkeypatch):
    """The RCCL part of pt_render_multi on a box with one GPU: with PT_MULTI_FORCE_RCCL=1 a single scene takes the
    general path, so librccl is loaded by name, a communicator is made for the device list and the grouped
    all-reduce runs (over one rank: the sum of one accumulator is that accumulator, bit for bit).  Twice, so that
    the second call finds the communicator the first one made."""
    import torch  # noqa: F401  (a process that already holds an RCCL shares it)
    ds = api.DeviceScene(util.world("atrium_small").export())
    cam = api.make_camera(vw=16, vh=9)
    a, ca = ds.render(cam, 160, 90, 6, seed=6)
    monkeypatch.setenv("PT_MULTI_FORCE_RCCL", "1")
    for _ in range(2):
        b, cb = api.render_multi([ds], cam, 160, 90, 6, seed=6)
        assert (util.bits(a) == util.bits(b)).all() and ca.as_dict() == cb.as_dict()
    ds.close()


def test_render_multi_over_all_devices(api):
    """The samples dealt over every visible device and merged by one RCCL all-reduce: the frame one device
    renders, up to the order of the float additions; ray counts add up exactly."""
    n = api.device_count()
    if n < 2:
        pytest.skip("one device")
    e = util.world("atrium_small").export()
    scenes = [api.DeviceScene(e, device=g) for g in range(n)]
    cam = api.make_camera(vw=16, vh=9)
    one, c1 = scenes[0].render(cam, 160, 90, 16, seed=6)
    many, cn = api.render_multi(scenes, cam, 160, 90, 16, seed=6)
    np.testing.assert_allclose(many, one, rtol=2e-5, atol=1e-6)
    assert (cn.paths, cn.rays_closest, cn.rays_shadow) == (c1.paths, c1.rays_closest, c1.rays_shadow)
    for s in scenes:
        s.close()


def test_current_device_is_left_alone(api):
    """Entry points that work on another device than the caller's current one restore the caller's."""
    import torch
    if api.device_count() < 2:
        pytest.skip("one device")
    torch.cuda.set_device(0)
    ds = api.DeviceScene(util.world("box").export(), device=1)
    ds.render(api.make_camera(vw=16, vh=9), 32, 18, 1)
    assert torch.cuda.current_device() == 0
    ds.close()
    assert torch.cuda.current_device() == 0


def test_counters_and_profile(api):
    ds = api.DeviceScene(util.world("box").export())
    cam = api.make_camera(vw=16, vh=9)
    ds.count_stats(True)
    ds.profile(True)
    _, c = ds.render(cam, 64, 36, 4, seed=1)
    p = ds.read_profile()
    assert c.node_boxes > c.rays_closest and c.tri_tests > 0 and c.shadow_node_boxes > 0
    # a frame this small is finished by the tail kernel from the first vertex on
    assert p["trace_launches"] + p["tail_launches"] >= 1 and p["trace_ms"] + p["shade_ms"] + p["tail_ms"] > 0
    assert p["trace_launches"] <= c.bounce_passes
    ds.count_stats(False)
    ds.profile(False)
    _, c2 = ds.render(cam, 64, 36, 4, seed=1)
    assert c2.node_boxes == 0 and (c2.rays_closest, c2.rays_shadow) == (c.rays_closest, c.rays_shadow)
    # the counting kernels and the plain ones render the same image
    a, _ = ds.render(cam, 64, 36, 4, seed=1)
    ds.count_stats(True)
    b, _ = ds.render(cam, 64, 36, 4, seed=1)
    assert (util.bits(a) == util.bits(b)).all()
    ds.close()


def test_profile_times_the_visibility_walk_of_split_shading(api, monkeypatch):
    """A scene outside on-chip memory shades in three kernels; the profile times the visibility walk on its own
    and counts its rays (the tail kernel switched off so that the wavefront passes do all the work)."""
    monkeypatch.setenv("PT_NO_RESIDENT", "1")
    monkeypatch.setenv("PT_TAIL", "0")
    ds = api.DeviceScene(util.world("terrain_small").export())
    cam = api.make_camera(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9)
    ds.profile(True)
    _, c = ds.render(cam, 96, 54, 2, seed=1)
    p = ds.read_profile()
    ds.profile(False)
    ds.close()
    assert p["visibility_launches"] == p["shade_launches"] >= 1
    assert 0 < p["visibility_ms"] <= p["shade_ms"]
    assert p["visibility_rays"] == c.rays_shadow > 0 and p["trace_rays"] == c.rays_closest


def test_exact_and_fast_agree_on_a_full_render(api):
    """Fast traversal only skips work that cannot change the answer: on a full multi-bounce render
    (millions of rays that start on surfaces) the two modes give the same radiance sums bit for bit."""
    ds = api.DeviceScene(util.world("atrium_small").export())
    cam = api.make_camera(vw=16, vh=9)
    a, ca = ds.render(cam, 160, 90, 8, seed=6, opt=api.make_options(trace_mode=api.PT_TRACE_EXACT))
    b, cb = ds.render(cam, 160, 90, 8, seed=6, opt=api.make_options(trace_mode=api.PT_TRACE_FAST))
    assert (util.bits(a) == util.bits(b)).all()
    assert (ca.rays_closest, ca.rays_shadow) == (cb.rays_closest, cb.rays_shadow)
    ds.close()


def test_headless_driver_renders_plane_obj(tmp_path):
    exe = os.path.join(ROOT, "pathtracer_amd", "lib", "ptrender")
    out = str(tmp_path / "plane.ppm")
    r = subprocess.run([exe, os.path.join(HERE, "golden", "plane.obj"), "-o", out, "--spp", "16"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "done" in r.stdout                       # camera.cpp:199
    lines = open(out).read().split("\n")
    assert lines[:3] == ["P3", "256 256", "255"]    # the reference's fixed size and spp (main.cpp:41-42)
    assert set(lines[3:3 + 256 * 256]) == {"0 0 0"}  # no emitter, black background


def test_host_pointer_calls_cross_their_staging_chunk():
    """pt_trace_closest / pt_trace_any / pt_trace_hits stage host rays through the device 2^24 at a
    time: the answers on either side of that boundary must be those of the same rays sent alone."""
    from pathtracer_amd import api
    from tests import util
    w = util.world("box")
    ds = api.DeviceScene(w.export())
    n = (1 << 24) + 777
    base_o, base_d = util.random_rays(4096, 5)
    reps = -(-n // 4096)
    o = np.tile(base_o, (reps, 1))[:n]
    d = np.tile(base_d, (reps, 1))[:n]
    small = ds.trace_closest(base_o, base_d)
    small_any = np.asarray(ds.trace_any(base_o, base_d))
    small_hits = ds.trace_hits(base_o, base_d)
    big = ds.trace_closest(o, d)
    big_any = np.asarray(ds.trace_any(o, d))
    big_hits = ds.trace_hits(o, d)
    ds.close()
    idx = np.arange(n) % 4096
    for lo, hi in ((0, 5000), ((1 << 24) - 3000, n)):
        s = slice(lo, hi)
        assert (big["tri"][s] == small["tri"][idx[s]]).all()
        assert (big["t"][s].view(np.uint32) == small["t"][idx[s]].view(np.uint32)).all()
        assert (big_any[s] == small_any[idx[s]]).all()
        assert (big_hits[s] == small_hits[idx[s]]).all()
    assert (big["tri"] == small["tri"][idx]).all()
