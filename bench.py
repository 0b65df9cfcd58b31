"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config box1k] [--spp S] [--scaling weak|strong]

One step = one frame: every pixel of the configuration's image gets its samples per pixel, each
camera path followed through all its bounces (closest-hit rays + visibility rays), and the
radiance sums are added to an accumulator that already lives in HBM.  With N > 1 every rank
renders sample indices rank, rank + N, ... of the same frame and one RCCL all-reduce merges the
accumulators at frame end; `--scaling weak` (default) gives every rank the configuration's full
sample count, `--scaling strong` splits the configuration's sample count over the ranks.

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
# Vector-instruction issue: 256 CUs x 4 SIMDs at 2.4 GHz.  The datasheet rate is one wave64 instruction
# per 2 cycles per SIMD.  What the chip sustains on the wall clock with every SIMD busy (8 waves per SIMD,
# profiles/r02_valu_rate.txt) is one per 4.5-4.8 nominal cycles for v_fma / v_pk_* / v_cndmask / v_cmp /
# v_min / v_max3 and one per 3.1-3.4 for v_mul / v_sub / v_mov: the clock drops under that load.
VALU_PEAK_DATASHEET_GINST = 1024 * 2.4 / 2
VALU_PEAK_SUSTAINED_GINST = 1024 * 2.4 / 4.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="box1k")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact", "fast-binary"],
                    help="fast-binary: the pruned walk over the reference-shaped trees, for comparison")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank renders the full sample count, strong = the sample count is split")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events around the kernels of the timed steps (kernel times and roofline become null)")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented pass (per-ray work counts become null)")
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
    """The oracle (a port of the reference's CPU renderer: recursive estimator, mt19937, rows handed to
    std::threads) timed on ALL of this box's hardware threads, as the reference uses them
    (threadcnt = std::thread::hardware_concurrency(), camera.cpp:49), on a bounded sample of the
    same workload."""
    from oracle import pt_oracle as O
    threads = O.hardware_threads()
    w = O.World.from_meshes(meshes, mats)
    cam = O.default_camera(**cam_kw)
    opt = O.default_options(**opt_kw)
    W, H = cfg["W"], cfg["H"]
    t0 = time.perf_counter()
    _, c1 = w.render(cam, W, H, 1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(64, round(12.0 / max(t1, 1e-3)))))
    t0 = time.perf_counter()
    _, c = w.render(cam, W, H, spp, spp_begin=1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    dt = time.perf_counter() - t0
    rays = c.rays_closest + c.rays_shadow
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%dx%d, %d spp of the same scene and camera on %d threads (%.1f s)" % (W, H, spp, threads, dt),
        "samples_per_s": c.paths / dt,
        "ref_semantics_node_reads_per_ray": c.node_boxes / max(1, rays),
        "ref_semantics_tri_tests_per_ray": c.tri_tests / max(1, rays),
    }


# Short sample counts of the secondary report lines (the other named configurations after the main run).
OTHERS_SPP = {"atrium260k": 32, "blob7m": 16, "terrain1m": 2}


def load_pmc(name):
    """Counter summary of this configuration's closest-hit kernel, kept under profiles/ by
    tools/profile_round.sh (rocprofv3 --pmc passes of this same benchmark command)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % name)
    if not os.path.exists(pmc):
        return {}
    with open(pmc) as f:
        return json.load(f)


def run_config(name, cfg, args, steps, warmup, count, rank, local_rank, world, dev):
    """Builds the scene, renders `warmup` untimed and `steps` timed frames of it and returns the
    measurements (on every rank; sums and the slowest rank's time are already reduced)."""
    import torch
    import torch.distributed as dist
    from pathtracer_amd import api, host

    W, H, spp_total = cfg["W"], cfg["H"], cfg["spp"]
    if args.scaling == "strong" and world > 1:
        # the configuration's samples, split: rank r renders indices r, r + world, ...
        from pathtracer_amd import distributed
        _, spp, _ = distributed.shard(spp_total, rank, world)
    else:
        spp = spp_total
    meshes, mats = cfg["scene"]()
    scene = host.Scene.from_meshes(meshes, mats)
    ds = scene.device_scene(local_rank)
    cam_kw = dict(cfg["cam"])
    if args.aperture >= 0:
        cam_kw.update(aperture=args.aperture, focus=3.0)
    cam = api.make_camera(**cam_kw)
    mode = args.mode
    opt_kw = dict(cfg.get("opt", {}))
    opt_kw["trace_mode"] = {"fast": api.PT_TRACE_FAST, "exact": api.PT_TRACE_EXACT, "fast-binary": api.PT_TRACE_FAST_BINARY}[mode]
    if args.max_depth >= 0:
        opt_kw["max_depth"] = args.max_depth
    if args.batch_spp > 0:
        opt_kw["batch_spp"] = args.batch_spp
    opt = api.make_options(spp_stride=world, **opt_kw)
    accum = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        accum.zero_()
        c = ds.render_dev(cam, W, H, spp, accum.data_ptr(), spp_begin=rank, seed=0, opt=opt, stream=stream)
        if world > 1:
            dist.all_reduce(accum)
        return c

    # instrumented pass: node boxes and triangles the traversal reads for exactly this frame (the
    # sample indices are the same, so the counts are those of every timed step)
    counts = None
    if count:
        ds.count_stats(True)
        counts = step()
        torch.cuda.synchronize()
        ds.count_stats(False)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ds.profile(not args.no_profile)
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
    info = ds.info
    if mode == "fast":
        resident = bool(112.0 * info["n_fast_nodes"] + 48.0 * info["n_tris"] + 32.0 * info["n_gates"] <= 90 * 1024)
    else:
        resident = bool(64.0 * info["n_wide_nodes"] + 48.0 * info["n_tris"] <= 60 * 1024)
    res = {
        "value": all_rays * K / dt / 1e6,
        "ms_per_step": dt / K * 1e3,
        "samples_per_s": all_paths * K / dt,
        "finite": bool(torch.isfinite(accum).all()),
        "workload": "%s: %d triangles in %d meshes, %d area light(s), %dx%d, %d spp per GPU%s, %s traversal"
                    % (name, info["n_tris"], info["n_meshes"], info["n_lights"], W, H, spp,
                       " (%d in all)" % spp_total if spp != spp_total else "", mode),
        "rays_per_sample": all_rays / max(1.0, all_paths),
        "bounce_passes": int(last.bounce_passes),
        "max_depth": opt.max_depth, "aperture": float(cam.aperture),
        "kernel_ms_per_step": {"trace_closest": prof["trace_ms"] / K, "shade_nee": prof["shade_ms"] / K,
                               "tail": prof["tail_ms"] / K, "raygen_resolve": prof["other_ms"] / K},
        "roofline": None,
    }
    launches = int(prof["trace_launches"])
    if launches > 0 and prof["trace_ms"] > 0:
        # The dominant kernel: k_trace_queue (closest hit).  Time: HIP events around every launch on the
        # launch stream, measured in this run.  HBM traffic and instruction counts per launch: the rocprofv3
        # counter passes of this same command kept under profiles/ (pmc_<config>.json).
        avg_ms = prof["trace_ms"] / launches
        pmc = load_pmc(name)
        if pmc.get("spp") not in (None, spp):
            pmc = {}  # collected at another sample count: the launches differ in size, per-launch figures do not apply
        traffic = pmc.get("trace_closest_hbm_bytes_per_launch")
        roof = {
            "bound": "hbm", "kernel": "k_trace_queue (closest hit)",
            "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
            "traffic": traffic,
            "avg_launch_ms": avg_ms, "launches": launches,
            "scene_lds_resident": resident,
            "l2_hit_rate": pmc.get("trace_closest_l2_hit_rate"),
            "profile": "profiles/pmc_%s.json" % name if pmc else None,
        }
        if traffic:
            # bytes that crossed the HBM side of L2, per second: the fraction of the HBM roofline this kernel uses
            roof["achieved"] = traffic / (avg_ms * 1e-3) / 1e9
            roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        if counts is not None:
            # algorithmic bytes of the traversal actually run: 28 B per child box tested in the fast tree (a
            # 112-byte node of four) or 32 B in the reference-shaped tree, 36 B per triangle tested, 53 B of
            # queue traffic per ray (32 B ray record in, 21 B hit record out).  All but the last are served by
            # LDS (a scene that fits) or by L2 / Infinity Cache (one that does not), so they are NOT priced
            # against the HBM peak; they say how much data the walk touches per ray.
            nb, nt, nr = counts.node_boxes, counts.tri_tests, counts.rays_closest
            roof["node_boxes_per_ray"] = nb / max(1, nr)
            roof["tri_tests_per_ray"] = nt / max(1, nr)
            roof["shadow_node_boxes_per_ray"] = counts.shadow_node_boxes / max(1, counts.rays_shadow)
            roof["shadow_tri_tests_per_ray"] = counts.shadow_tri_tests / max(1, counts.rays_shadow)
            box_bytes = 28.0 if mode == "fast" else 32.0
            roof["algorithmic_bytes_per_ray"] = (box_bytes * nb + 36.0 * nt + 53.0 * nr) / max(1, nr)
            roof["algorithmic_hbm_bytes_per_ray"] = 53.0 if resident else None
        valu = pmc.get("trace_closest_valu_insts_per_launch")
        if valu:
            # what binds a scene that lives in LDS: wave-level vector instructions per second
            rate = valu / (avg_ms * 1e-3) / 1e9
            roof["valu_issue"] = {
                "achieved": rate, "unit": "G wave-instr/s",
                "peak_datasheet": VALU_PEAK_DATASHEET_GINST, "frac_datasheet": rate / VALU_PEAK_DATASHEET_GINST,
                "peak_sustained": VALU_PEAK_SUSTAINED_GINST, "frac_sustained": rate / VALU_PEAK_SUSTAINED_GINST,
                "active_lanes_of_64": pmc.get("trace_closest_active_lanes"),
                "lds_bank_conflict_frac": pmc.get("trace_closest_lds_conflict_frac"),
            }
            if resident:
                roof["binding_bound"] = "valu_issue"
        res["roofline"] = roof
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
    res, (meshes, mats, cam_kw, opt_kw) = run_config(args.config, cfg, args, args.steps, args.warmup, not args.no_count,
                                                     rank, local_rank, world, dev)
    out = {
        "metric": "Mrays/s at 1080p" if cfg["H"] == 1080 else "Mrays/s",
        "value": res["value"], "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
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

    # the other named configurations, one short frame each: same sharding over the ranks, same
    # counters; these are report lines, `value` above is the metric
    if args.others and args.config == "box1k":
        others = []
        for name, spp in OTHERS_SPP.items():
            c = dict(scenes.CONFIGS[name])
            c["spp"] = spp
            r, _ = run_config(name, c, args, 1, 1, not args.no_count, rank, local_rank, world, dev)
            ro = r["roofline"] or {}
            others.append({
                "workload": r["workload"], "value": r["value"], "unit": "Mrays/s", "ms_per_step": r["ms_per_step"],
                "samples_per_s": r["samples_per_s"], "rays_per_sample": r["rays_per_sample"],
                "max_depth": r["max_depth"], "aperture": r["aperture"],
                "kernel_ms_per_step": r["kernel_ms_per_step"], "finite": r["finite"],
                "hbm_roofline_frac": ro.get("frac"), "hbm_traffic_per_launch": ro.get("traffic"),
                "l2_hit_rate": ro.get("l2_hit_rate"),
                "node_boxes_per_ray": ro.get("node_boxes_per_ray"), "tri_tests_per_ray": ro.get("tri_tests_per_ray"),
            })
        out["other_configs"] = others
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
