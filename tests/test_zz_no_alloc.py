"""pt_render_dev_static — the entry point that allocates nothing — against pt_render_dev, through the
C ABI: same radiance sums bit for bit and the same counters, on two calls in a row, the second with a
smaller input (the static buffer keeps the first call's contents); device memory in use does not move
across the calls; a frame beyond the buffer is rejected with a clear error."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from pathtracer_amd import api
    assert api.device_count() >= 1
    return api


def _pair(api, ds, cam, W, H, spp, seed, opt, torch):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    b = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    # the kernels' code may be loaded at their first launch: let a tiny frame do that first
    warm = torch.zeros(8 * 8 * 3, dtype=torch.float32, device=dev)
    ds.render_dev_static(cam, 8, 8, 1, warm.data_ptr(), seed=seed, opt=opt, stream=st)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    cb = ds.render_dev_static(cam, W, H, spp, b.data_ptr(), seed=seed, opt=opt, stream=st)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free1 == free0, "device memory in use changed across pt_render_dev_static (%d -> %d)" % (free0, free1)
    ca = ds.render_dev(cam, W, H, spp, a.data_ptr(), seed=seed, opt=opt, stream=st)
    torch.cuda.synchronize()
    return a.cpu().numpy(), ca, b.cpu().numpy(), cb


@pytest.mark.parametrize("name,cam_kw,opt_kw", [
    ("box", dict(vw=16, vh=9), dict()),                                            # scene in LDS: fused shading
    ("atrium_small", dict(vw=16, vh=9, aperture=0.05, focus=3.0), dict(max_depth=5)),  # split pipeline, one light
    ("terrain_small", dict(origin=(0, 0.5, 0), euler=(0.15, 0, 0), vw=16, vh=9), dict()),  # split pipeline, 20 lights
])
def test_static_entry_point_equals_render_dev_twice_in_a_row(api, name, cam_kw, opt_kw):
    import torch
    ds = api.DeviceScene(util.world(name).export())
    cam = api.make_camera(**cam_kw)
    opt = api.make_options(**opt_kw)
    for W, H, spp, seed in ((320, 180, 5, 3), (128, 72, 3, 4)):   # the second call is the smaller one
        a, ca, b, cb = _pair(api, ds, cam, W, H, spp, seed, opt, torch)
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), "%s %dx%d: radiance sums differ" % (name, W, H)
        da, db = ca.as_dict(), cb.as_dict()
        da.pop("bounce_passes"), db.pop("bounce_passes")   # kernel passes: depends on how many samples fit at once
        assert da == db
        assert a.sum() > 0
    ds.close()


def test_static_entry_point_accumulates_and_shards_like_render_dev(api):
    """It ADDS to the accumulator and honours spp_begin / spp_stride like the entry point it mirrors."""
    import torch
    ds = api.DeviceScene(util.world("box").export())
    cam = api.make_camera(vw=16, vh=9)
    dev = torch.device("cuda", 0)
    W, H = 160, 90
    st = torch.cuda.current_stream().cuda_stream
    a = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    b = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    for begin in (0, 1):
        opt = api.make_options(spp_stride=2)
        ds.render_dev(cam, W, H, 3, a.data_ptr(), spp_begin=begin, seed=9, opt=opt, stream=st)
        ds.render_dev_static(cam, W, H, 3, b.data_ptr(), spp_begin=begin, seed=9, opt=opt, stream=st)
    torch.cuda.synchronize()
    assert (a.cpu().numpy().view(np.uint32) == b.cpu().numpy().view(np.uint32)).all()
    ds.close()


def test_static_entry_point_rejects_frames_beyond_its_buffer(api):
    import torch
    ds = api.DeviceScene(util.world("box").export())
    cam = api.make_camera(vw=16, vh=9)
    W, H = 1920, 1080   # 2 M pixels x ~150 B > 64 MiB
    acc = torch.zeros(H * W * 3, dtype=torch.float32, device=torch.device("cuda", 0))
    with pytest.raises(api.PtError) as ei:
        ds.render_dev_static(cam, W, H, 1, acc.data_ptr())
    assert ei.value.code == -4 and "static buffer" in str(ei.value)
    assert api.lib().pt_static_scratch_bytes() == 64 << 20
    torch.cuda.synchronize()
    assert float(acc.sum()) == 0.0
    # the limit the library states is the limit it enforces
    cap = ds.static_max_pixels()
    assert 300_000 < cap < 400_000
    small = torch.zeros(3 * cap, dtype=torch.float32, device=torch.device("cuda", 0))
    side = int(cap ** 0.5)
    ds.render_dev_static(cam, side, side, 1, small.data_ptr())              # just inside
    with pytest.raises(api.PtError):
        ds.render_dev_static(cam, cap + 64, 1, 1, small.data_ptr())         # just outside
    ds.close()


def test_static_calls_from_two_scenes_on_two_streams_are_ordered(api):
    """The static buffer is one per GPU.  Calls that name different scenes and different streams are ordered
    on the device by the library (an event per GPU), so each still equals pt_render_dev bit for bit."""
    import torch
    dev = torch.device("cuda", 0)
    names = ("box", "atrium_small")
    scenes_ = [api.DeviceScene(util.world(n).export()) for n in names]
    cam = api.make_camera(vw=16, vh=9)
    W, H, spp = 256, 144, 3
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    want = []
    for ds in scenes_:
        a = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        ds.render_dev(cam, W, H, spp, a.data_ptr(), seed=5)
        torch.cuda.synchronize()
        want.append(a.cpu().numpy())
    outs = [torch.zeros(H * W * 3, dtype=torch.float32, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    for k in range(4):            # box, atrium, box, atrium: alternating scenes and streams, no host wait between
        scenes_[k % 2].render_dev_static(cam, W, H, spp, outs[k].data_ptr(), seed=5, stream=streams[k % 2].cuda_stream)
    torch.cuda.synchronize()
    for k in range(4):
        assert (outs[k].cpu().numpy().view(np.uint32) == want[k % 2].view(np.uint32)).all(), k
    for ds in scenes_:
        ds.close()
