"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config box1k] [--spp S]

One step = one frame: every pixel of the configuration's image gets `spp` camera paths,
each followed through all its bounces (closest-hit rays + visibility rays), and the
radiance sums are added to an accumulator that already lives in HBM.  With N > 1 each
rank renders its own `spp` sample indices of the same frame (global index = rank + k*N,
weak scaling) and one RCCL all-reduce merges the accumulators at frame end.

Rank 0 prints ONE JSON line.  `value` is rays per second over all ranks (a ray = one
Scene::castray-equivalent query: primary, bounce or visibility).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
VALU_PEAK_GINST = 1024 * 2.4 / 4  # wave64 vector instructions per ns: 256 CUs x 4 SIMDs, 4 cycles each, 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="box1k")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact", "fast-binary"],
                    help="fast-binary: the pruned walk over the reference-shaped trees, for comparison")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented pass (roofline becomes null)")
    ap.add_argument("--max-depth", type=int, default=-1)
    ap.add_argument("--aperture", type=float, default=-1.0)
    ap.add_argument("--others", dest="others", action="store_true", default=True,
                    help="after the main run, one short frame of each other named configuration (default)")
    ap.add_argument("--no-others", dest="others", action="store_false")
    ap.add_argument("--batch-spp", type=int, default=0, help="samples per pixel in flight together (0 = library default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the N>1 path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 (rehearsal of the N>1 path on a one-GPU box, with --backend gloo)")
    return ap.parse_args()


def workload(args):
    from pathtracer_amd import scenes
    if args.config not in scenes.CONFIGS or scenes.CONFIGS[args.config]["scene"] is None:
        raise SystemExit("unknown or non-benchmark config %r" % args.config)
    cfg = dict(scenes.CONFIGS[args.config])
    if args.spp > 0:
        cfg["spp"] = args.spp
    return cfg


def cpu_baseline(meshes, mats, cfg, cam_kw, opt_kw):
    """The oracle (a port of the reference's CPU renderer: recursive estimator, mt19937,
    rows handed to std::threads) timed on this box's host cores on a bounded sample of the
    same workload."""
    from oracle import pt_oracle as O
    threads = min(16, O.hardware_threads())
    w = O.World.from_meshes(meshes, mats)
    cam = O.default_camera(**cam_kw)
    opt = O.default_options(**opt_kw)
    W, H = cfg["W"], cfg["H"]
    t0 = time.perf_counter()
    _, c1 = w.render(cam, W, H, 1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(16, round(12.0 / max(t1, 1e-3)))))
    t0 = time.perf_counter()
    _, c = w.render(cam, W, H, spp, spp_begin=1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    dt = time.perf_counter() - t0
    rays = c.rays_closest + c.rays_shadow
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%dx%d, %d spp of the same scene and camera (%.1f s)" % (W, H, spp, dt),
        "samples_per_s": c.paths / dt,
        "ref_semantics_node_reads_per_ray": c.node_boxes / max(1, rays),
        "ref_semantics_tri_tests_per_ray": c.tri_tests / max(1, rays),
    }


# Named settings of the larger configurations: the 260 k-triangle one is quoted with 5 bounces and
# depth of field on.  `spp` here is the short sample count of the secondary report lines.
OTHERS = {
    "atrium260k": dict(spp=32, max_depth=5, aperture=0.05),
    "blob7m": dict(spp=16),
    "terrain1m": dict(spp=2),
}


def run_config(name, cfg, mode, max_depth, aperture, steps, warmup, count, rank, local_rank, world, dev, batch_spp=0):
    """Builds the scene, renders `warmup` untimed and `steps` timed frames of it and returns the
    measurements (on every rank; sums and the slowest rank's time are already reduced)."""
    import torch
    import torch.distributed as dist
    from pathtracer_amd import api, host

    W, H, spp = cfg["W"], cfg["H"], cfg["spp"]
    meshes, mats = cfg["scene"]()
    scene = host.Scene.from_meshes(meshes, mats)
    ds = scene.device_scene(local_rank)
    cam_kw = dict(cfg["cam"])
    if aperture >= 0:
        cam_kw.update(aperture=aperture, focus=3.0)
    cam = api.make_camera(**cam_kw)
    opt_kw = dict(trace_mode={"fast": api.PT_TRACE_FAST, "exact": api.PT_TRACE_EXACT, "fast-binary": api.PT_TRACE_FAST_BINARY}[mode])
    if max_depth >= 0:
        opt_kw["max_depth"] = max_depth
    if batch_spp > 0:
        opt_kw["batch_spp"] = batch_spp
    opt = api.make_options(spp_stride=world, **opt_kw)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        accum.zero_()
        c = ds.render_dev(cam, W, H, spp, accum.data_ptr(), spp_begin=rank, seed=0, opt=opt, stream=stream)
        if world > 1:
            dist.all_reduce(accum)
        return c

    # instrumented pass: node boxes and triangles the closest-hit kernel reads for exactly this
    # frame (the sample indices are the same, so the counts are those of every timed step)
    counts = None
    if count:
        ds.count_stats(True)
        counts = step()
        torch.cuda.synchronize()
        ds.count_stats(False)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ds.profile(True)
    ds.read_profile()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ds.read_profile()
    ds.profile(False)

    rays = last.rays_closest + last.rays_shadow
    tot = torch.tensor([dt, float(rays), float(last.paths)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    all_rays, all_paths = float(tot[1]), float(tot[2])
    K = max(1, steps)
    res = {
        "value": all_rays * K / dt / 1e6,
        "ms_per_step": dt / K * 1e3,
        "samples_per_s": all_paths * K / dt,
        "finite": bool(torch.isfinite(accum).all()),
        "workload": "%s: %d triangles in %d meshes, %d area light(s), %dx%d, %d spp per GPU, %s traversal"
                    % (name, ds.info["n_tris"], ds.info["n_meshes"], ds.info["n_lights"], W, H, spp, mode),
        "rays_per_sample": all_rays / max(1.0, all_paths),
        "bounce_passes": int(last.bounce_passes),
        "max_depth": opt.max_depth, "aperture": float(cam.aperture),
        "kernel_ms_per_step": {"trace_closest": prof["trace_ms"] / K, "shade_nee": prof["shade_ms"] / K,
                               "raygen_resolve": prof["other_ms"] / K},
        "roofline": None,
    }
    if counts is not None:
        # closest-hit kernel: algorithmic bytes = 32 B per node box + 36 B per triangle + 48 B of
        # ray in / hit out per ray, over all launches of the timed steps
        nb, nt, nr = counts.node_boxes, counts.tri_tests, counts.rays_closest
        bytes_step = 32.0 * nb + 36.0 * nt + 48.0 * nr
        launches = max(1, int(prof["trace_launches"]))
        gbs = bytes_step * K / (prof["trace_ms"] * 1e-3) / 1e9 if prof["trace_ms"] > 0 else 0.0
        traffic = valu_insts = lanes = None
        pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % name)
        if os.path.exists(pmc):
            with open(pmc) as f:
                rec = json.load(f)
            traffic = rec.get("trace_closest_hbm_bytes_per_launch")
            valu_insts = rec.get("trace_closest_valu_insts_per_launch")
            lanes = rec.get("trace_closest_active_lanes")
        scene_bytes = 64.0 * ds.info["n_wide_nodes"] + 48.0 * ds.info["n_tris"]
        res["roofline"] = {
            "bound": "hbm", "kernel": "k_trace_queue (closest hit)",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_step * K / launches,
            "avg_launch_ms": prof["trace_ms"] / launches, "launches": launches,
            "node_boxes_per_ray": nb / max(1, nr), "tri_tests_per_ray": nt / max(1, nr),
            "shadow_node_boxes_per_ray": counts.shadow_node_boxes / max(1, counts.rays_shadow),
            "shadow_tri_tests_per_ray": counts.shadow_tri_tests / max(1, counts.rays_shadow),
            # the gathers counted above are served by the copy of the scene each workgroup keeps in
            # LDS when it fits (then frac can exceed 1 and the kernel is bound by vector-instruction
            # issue); only the ray queue and the hit records (44 B per ray) must cross HBM
            "scene_bytes": scene_bytes,
            "scene_lds_resident": bool(scene_bytes <= 150 * 1024),
            "queue_bytes_gbs": 44.0 * nr * K / (prof["trace_ms"] * 1e-3) / 1e9 if prof["trace_ms"] > 0 else 0.0,
        }
        if valu_insts and prof["trace_ms"] > 0:
            # what does bound an LDS-resident scene: wave-level vector instructions per second against
            # the issue peak of 1024 SIMDs at one wave64 instruction per 4 cycles of 2.4 GHz; counts
            # per launch from the rocprofv3 counter pass kept under profiles/, time measured here
            rate = valu_insts / (prof["trace_ms"] * 1e-3 / launches)
            res["roofline"]["valu_issue"] = {"achieved": rate / 1e9, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                                            "frac": rate / 1e9 / VALU_PEAK_GINST, "active_lanes_of_64": lanes}
    ds.close()
    scene.close()
    return res, (meshes, mats, cam_kw, {k: v for k, v in opt_kw.items() if k not in ("trace_mode", "batch_spp")})


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    from pathtracer_amd import scenes

    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg = workload(args)
    res, (meshes, mats, cam_kw, opt_kw) = run_config(args.config, cfg, args.mode, args.max_depth, args.aperture,
                                                     args.steps, args.warmup, not args.no_count, rank, local_rank,
                                                     world, dev, args.batch_spp)
    out = {
        "metric": "Mrays/s at 1080p" if cfg["H"] == 1080 else "Mrays/s",
        "value": res["value"], "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": res["workload"], "rays_per_sample": res["rays_per_sample"],
                   "bounce_passes": res["bounce_passes"], "max_depth": res["max_depth"], "aperture": res["aperture"]},
        "samples_per_s": res["samples_per_s"], "finite": res["finite"],
        "kernel_ms_per_step": res["kernel_ms_per_step"],
    }
    if res["roofline"] is not None:
        out["roofline"] = res["roofline"]
    if world == 1 and rank == 0 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(meshes, mats, cfg, cam_kw, opt_kw)
    del meshes, mats

    # the other named configurations, one short frame each (rates do not depend on spp): same
    # sharding over the ranks, same counters; these are report lines, `value` above is the metric
    if args.others and args.config == "box1k":
        others = []
        for name, kw in OTHERS.items():
            c = dict(scenes.CONFIGS[name])
            c["spp"] = kw["spp"]
            r, _ = run_config(name, c, args.mode, kw.get("max_depth", -1), kw.get("aperture", -1.0), 1, 1,
                              not args.no_count, rank, local_rank, world, dev)
            others.append({
                "workload": r["workload"], "value": r["value"], "unit": "Mrays/s", "ms_per_step": r["ms_per_step"],
                "samples_per_s": r["samples_per_s"], "max_depth": r["max_depth"], "aperture": r["aperture"],
                "kernel_ms_per_step": r["kernel_ms_per_step"], "finite": r["finite"],
                "roofline_frac": None if r["roofline"] is None else r["roofline"]["frac"],
                "node_boxes_per_ray": None if r["roofline"] is None else r["roofline"]["node_boxes_per_ray"],
                "tri_tests_per_ray": None if r["roofline"] is None else r["roofline"]["tri_tests_per_ray"],
            })
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
