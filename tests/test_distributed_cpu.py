"""The N > 1 path on CPU: two gloo ranks each render their strided share of the samples and one
all-reduce merges the radiance sums (pathtracer_amd/distributed.py).  The per-rank renderer here is
the oracle (the GPU library cannot run without a GPU); what is under test is the sharding and the
reduction, which are the same code the GPU path uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import distributed


def test_shard_covers_every_sample_once():
    for total in (0, 1, 2, 7, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, c, s = distributed.shard(total, r, world)
                seen += [b + k * s for k in range(c)]
            assert sorted(seen) == list(range(total))
    with pytest.raises(ValueError):
        distributed.shard(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pt_oracle as O
    from pathtracer_amd import scenes
    w = O.World.from_meshes(*scenes.box_room(6))
    cam = O.default_camera(vw=16, vh=9)
    W, H, spp = 48, 27, 6

    def render_fn(begin, count, stride):
        img, _ = w.render(cam, W, H, count, spp_begin=begin, seed=9, opt=O.default_options(spp_stride=stride),
                          threads=1)
        return torch.from_numpy(img.reshape(-1).copy())

    got = distributed.render_sharded(render_fn, spp)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sum_to_the_single_rank_frame(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from oracle import pt_oracle as O
    from pathtracer_amd import scenes
    w = O.World.from_meshes(*scenes.box_room(6))
    full, _ = w.render(O.default_camera(vw=16, vh=9), 48, 27, 6, seed=9, threads=1)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert (r0 == r1).all()                      # all-reduce: every rank holds the frame
    np.testing.assert_allclose(r0, full.reshape(-1), rtol=2e-6, atol=1e-7)   # float sum reordered
