"""Sample sharding over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

Samples are independent and the random numbers of a sample depend only on (seed, pixel,
global sample index), so rank r of G renders the global indices r, r+G, r+2G, ... of every
pixel and ONE all-reduce of the W*H*3 float32 radiance sums at frame end gives every rank
the full frame (24.9 MB at 1080p).  Nothing is exchanged during the frame.
"""


def shard(spp_total, rank, world):
    """(spp_begin, spp_count, spp_stride) of rank `rank` out of `world` for spp_total samples."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    count = (spp_total - rank + world - 1) // world if spp_total > rank else 0
    return rank, count, world


def render_sharded(render_fn, spp_total, rank=None, world=None, group=None):
    """render_fn(spp_begin, spp_count, spp_stride) -> torch tensor of radiance sums on this
    rank's device.  Returns the tensor summed over all ranks (in place)."""
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, count, stride = shard(spp_total, rank, world)
    accum = render_fn(begin, count, stride)
    if world > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
    return accum


def render_frame(device_scene, cam, W, H, spp_total, seed=0, opt_kwargs=None, group=None):
    """Renders spp_total samples per pixel split over the ranks of the default (or given) process
    group; every rank returns the full [H, W, 3] tensor of radiance sums on its own GPU."""
    import torch
    from . import api

    dev = torch.device("cuda", device_scene.device)

    def render_fn(begin, count, stride):
        accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
        opt = api.make_options(spp_stride=stride, **(opt_kwargs or {}))
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            device_scene.render_dev(cam, W, H, count, accum.data_ptr(), spp_begin=begin, seed=seed, opt=opt,
                                    stream=stream)
        return accum

    return render_sharded(render_fn, spp_total, group=group).view(H, W, 3)
