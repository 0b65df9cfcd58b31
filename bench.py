"""Benchmark of the render hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config box1k] [--spp S] [--scaling weak|strong]
                  [--multi processes|inprocess] [--no-others] [--no-cpu]

One step = one frame: every pixel of the configuration's image gets its samples per pixel, each
camera path followed through all its bounces (closest-hit rays + visibility rays), and the
radiance sums are added to an accumulator that already lives in HBM.

N > 1, `--multi processes` (default): one process per GPU.  Under a launcher (torch.distributed.run sets
WORLD_SIZE) this process is one rank; without one, this process starts the N ranks itself as child
processes before it touches a GPU and relays rank 0's line.  Every rank renders sample indices rank,
rank + N, ... of the same frame and one RCCL all-reduce of the W*H*3 float accumulator merges them at frame
end.  `--scaling weak` gives every rank the configuration's full sample count, `--scaling strong` splits a
fixed total over the ranks; left out, each configuration uses the scaling it is named with (the 7 M-triangle
and the 4K many-light configurations are strong, the others weak).
N > 1, `--multi inprocess`: one process, one host thread per device, pt_render_multi (RCCL loaded by the
library itself); the accumulator is handed back in host memory, so this form includes that copy.

Rank 0 prints ONE JSON line.  `value` is rays per second over all ranks (a ray = one
Scene::castray-equivalent query: primary, bounce or visibility) on the main configuration; after it every
other named configuration is measured the same way (`configs`), each with its own roofline block and CPU
baseline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
# Vector-instruction issue: 256 CUs x 4 SIMDs at 2.4 GHz, one wave64 instruction per 2 cycles per SIMD (the
# datasheet rate, which every fraction below is quoted against).  With every SIMD busy the chip holds about
# half its nominal clock (tools/ubench/valu_rate.hip: 2.0-2.6 shader cycles per wave instruction by s_memtime,
# 4.5-4.8 nominal cycles by the wall clock); that sustained rate is reported beside it, never instead of it.
VALU_PEAK_DATASHEET_GINST = 1024 * 2.4 / 2
VALU_SUSTAINED_GINST = 1024 * 2.4 / 4.6

# Samples per pixel of the benchmark lines.  N = 1: the named sample count where a frame stays within seconds
# (box1k 64, atrium260k 256), 16 for the 7 M triangles, 64 of the 4K frame's 4 096.  N > 1 with strong scaling:
# a fixed total that leaves every GPU of an 8-GPU node the N = 1 line's work (128 = 8 x 16, 512 = 8 x 64, the
# per-GPU share BASELINE config 5 names).
BENCH_SPP = {"box1k": 64, "atrium260k": 256, "blob7m": 16, "terrain1m": 64}
STRONG_TOTAL_SPP = {"box1k": 64, "atrium260k": 256, "blob7m": 128, "terrain1m": 512}
NAMED_SCALING = {"box1k": "weak", "atrium260k": "weak", "blob7m": "strong", "terrain1m": "strong"}
OTHERS = ["atrium260k", "blob7m", "terrain1m"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="box1k")
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact", "fast-binary"],
                    help="fast-binary: the pruned walk over the reference-shaped trees, for comparison")
    ap.add_argument("--scaling", default="", choices=["", "weak", "strong"],
                    help="N > 1: weak = every rank renders the full sample count, strong = a fixed total is split; "
                         "default: what each configuration is named with")
    ap.add_argument("--multi", default="processes", choices=["processes", "inprocess"],
                    help="N > 1: one process per GPU (torch.distributed), or one process driving pt_render_multi")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events around the kernels of the timed steps (kernel times and roofline become null)")
    ap.add_argument("--no-count", action="store_true", help="skip the instrumented pass (per-ray work counts become null)")
    ap.add_argument("--max-depth", type=int, default=-1)
    ap.add_argument("--aperture", type=float, default=-1.0)
    ap.add_argument("--others", dest="others", action="store_true", default=True,
                    help="after the main run, the other named configurations (default)")
    ap.add_argument("--no-others", dest="others", action="store_false")
    ap.add_argument("--other-steps", type=int, default=3, help="timed frames of each other configuration")
    ap.add_argument("--batch-spp", type=int, default=0, help="samples per pixel in flight together (0 = library default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo only to rehearse the N>1 path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank / every scene copy uses device 0 (rehearsal of the N>1 paths on a one-GPU box: "
                         "with --backend gloo for processes; in-process the merge then runs without RCCL)")
    return ap.parse_args()


def config_of(name, args):
    from pathtracer_amd import scenes
    if name not in scenes.CONFIGS or scenes.CONFIGS[name]["scene"] is None:
        raise SystemExit("unknown or non-benchmark config %r" % name)
    cfg = dict(scenes.CONFIGS[name])
    cfg["named_spp"] = cfg["spp"]
    return cfg


def samples_for(name, args, world, main):
    """(samples this run renders in all, scaling label) for a configuration on `world` GPUs."""
    scaling = args.scaling or NAMED_SCALING.get(name, "weak")
    if main and args.spp > 0:
        base = args.spp
        return (base if scaling == "strong" or world == 1 else base * world), scaling
    if world == 1:
        return BENCH_SPP[name], scaling
    if scaling == "strong":
        return STRONG_TOTAL_SPP[name], scaling
    return BENCH_SPP[name] * world, scaling


def cpu_baseline(meshes, mats, cfg, cam_kw, opt_kw, budget_s=12.0):
    """The oracle (a port of the reference's CPU renderer: recursive estimator, mt19937, rows handed to
    std::threads) timed on ALL of this box's hardware threads, as the reference uses them
    (threadcnt = std::thread::hardware_concurrency(), camera.cpp:49), on a bounded sample of the same workload:
    the same scene, camera and integrator settings, the frame reduced by an integer factor where one sample of
    the full frame would not fit the time box, then as many samples as fit in about `budget_s` seconds."""
    from oracle import pt_oracle as O
    threads = O.hardware_threads()
    w = O.World.from_meshes(meshes, mats)
    cam = O.default_camera(**cam_kw)
    opt = O.default_options(**opt_kw)
    W, H = cfg["W"], cfg["H"]
    # probe: one sample of the frame at 1/8 of its width and height
    pw, ph = max(16, W // 8), max(9, H // 8)
    t0 = time.perf_counter()
    _, c0 = w.render(cam, pw, ph, 1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    t_px = max(time.perf_counter() - t0, 1e-4) / (pw * ph)          # seconds per path
    reduce_by = 1
    while reduce_by < 8 and t_px * (W // reduce_by) * (H // reduce_by) > budget_s / 2:
        reduce_by *= 2
    sw, sh = W // reduce_by, H // reduce_by
    spp = int(max(1, min(64, round(budget_s / (t_px * sw * sh)))))
    t0 = time.perf_counter()
    _, c = w.render(cam, sw, sh, spp, spp_begin=1, seed=1, opt=opt, rng=1, form=0, threads=threads)
    dt = time.perf_counter() - t0
    rays = c.rays_closest + c.rays_shadow
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%dx%d%s, %d spp of the same scene, camera and integrator settings on %d threads (%.1f s)"
                  % (sw, sh, " (the %dx%d frame at 1/%d)" % (W, H, reduce_by) if reduce_by > 1 else "", spp, threads, dt),
        "samples_per_s": c.paths / dt,
        "ref_semantics_node_reads_per_ray": c.node_boxes / max(1, rays),
        "ref_semantics_tri_tests_per_ray": c.tri_tests / max(1, rays),
    }


def load_pmc(name):
    """Counter summary of this configuration's closest-hit kernel, kept under profiles/ by
    tools/profile_round.sh (rocprofv3 --pmc passes of this same benchmark command)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % name)
    if not os.path.exists(pmc):
        return {}
    with open(pmc) as f:
        return json.load(f)


def roofline_block(name, mode, spp, info, prof, counts, count_spp):
    """Roofline of the dominant kernel, k_trace_queue (closest hit).  Time: HIP events around every launch on
    the launch stream, measured in this run.  `achieved`: the algorithmic bytes of the traversal actually run
    (28 B per child box tested in the fast tree's 112-byte nodes of four, 10 B in its 80-byte nodes of eight, or 32 B
    in the reference-shaped tree;
    36 B per triangle tested; 53 B of queue traffic per ray: 32 B ray record in, 21 B hit record out; the box
    and triangle counts come from the instrumented pass of this run) per second.  For a scene that lives in LDS
    only the 53 B per ray can come from HBM, and that is what is priced.  `traffic`: HBM-side bytes per launch
    from the rocprofv3 counter passes of this same command (profiles/pmc_<config>.json, request sizes resolved
    as tools/pmc_to_traffic.py describes)."""
    launches = int(prof["trace_launches"])
    if launches <= 0 or prof["trace_ms"] <= 0:
        return None
    avg_ms = prof["trace_ms"] / launches
    pmc = load_pmc(name)
    if pmc.get("spp") not in (None, spp):
        pmc = {}  # collected at another sample count: the launches differ in size, per-launch figures do not apply
    if mode == "fast":
        resident = bool(112.0 * info["n_fast_nodes"] + 48.0 * info["n_tris"] + 32.0 * info["n_gates"] <= 90 * 1024)
    else:
        resident = bool(64.0 * info["n_wide_nodes"] + 48.0 * info["n_tris"] <= 60 * 1024)
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
        roof["hbm_side"] = {"bytes_per_launch": traffic, "GB_per_s": traffic / (avg_ms * 1e-3) / 1e9,
                            "frac": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "how": pmc.get("hbm_bytes_how")}
    if counts is not None and counts.rays_closest:
        nb, nt, nr = counts.node_boxes, counts.tri_tests, counts.rays_closest
        # per child box: 28 B in the four-child tree (112-byte node), 10 B in the eight-child tree (80-byte node)
        oct_tree = mode == "fast" and not resident and info.get("n_oct_nodes", 0) > 0
        box_bytes = (10.0 if oct_tree else 28.0) if mode == "fast" else 32.0
        roof["tree"] = "eight-child, boxes on an 8-bit grid" if oct_tree else ("four-child" if mode == "fast" else "reference-shaped")
        per_ray = (box_bytes * nb + 36.0 * nt) / nr + 53.0
        roof["node_boxes_per_ray"] = nb / nr
        roof["tri_tests_per_ray"] = nt / nr
        roof["shadow_node_boxes_per_ray"] = counts.shadow_node_boxes / max(1, counts.rays_shadow)
        roof["shadow_tri_tests_per_ray"] = counts.shadow_tri_tests / max(1, counts.rays_shadow)
        roof["algorithmic_bytes_per_ray"] = per_ray
        roof["algorithmic_hbm_bytes_per_ray"] = 53.0 if resident else per_ray
        roof["counted_at_spp"] = count_spp
    rays_per_launch = prof.get("trace_rays", 0) / launches if prof.get("trace_rays") else None
    if rays_per_launch and roof.get("algorithmic_hbm_bytes_per_ray"):
        roof["rays_per_launch"] = rays_per_launch
        roof["achieved"] = roof["algorithmic_hbm_bytes_per_ray"] * rays_per_launch / (avg_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    valu = pmc.get("trace_closest_valu_insts_per_launch")
    if valu:
        rate = valu / (avg_ms * 1e-3) / 1e9
        roof["valu_issue"] = {
            "achieved": rate, "unit": "G wave-instr/s",
            "peak_datasheet": VALU_PEAK_DATASHEET_GINST, "frac_datasheet": rate / VALU_PEAK_DATASHEET_GINST,
            "sustained_under_load": VALU_SUSTAINED_GINST,
            "active_lanes_of_64": pmc.get("trace_closest_active_lanes"),
            "lds_bank_conflict_frac": pmc.get("trace_closest_lds_conflict_frac"),
            "wave_time_waiting_on_memory": pmc.get("trace_closest_wait_any_frac"),
        }
    # what binds the kernel: a scene in LDS issues vector instructions back to back; a scene outside it waits
    # for its scattered 16-byte gathers (a 112-byte node per lane per step) long before HBM bandwidth runs out
    roof["binding_bound"] = "valu_issue" if resident else "scattered_gather_rate (L1/L2 requests per second), not HBM bytes"
    if not resident and roof.get("frac") is not None:
        roof["note"] = ("the scene is served by L2 and the Infinity Cache as well as HBM, so the algorithmic bytes per "
                        "second may exceed the HBM peak; hbm_side is what crossed the memory side of L2")
    return roof


def visibility_block(name, spp, info, prof, counts):
    """The same figures for k_shadow_walk, the visibility walk of split shading (scenes outside LDS; a scene in
    LDS walks its visibility rays inside the shading kernel).  Algorithmic bytes per ray: 29 B of stream (hit
    point 12 B, direction 12 B, the vertex's t 4 B in, the answer 1 B out) + 10 B per child box of the
    eight-child tree (28 B in the four-child tree) + 36 B per triangle tested."""
    launches = int(prof.get("visibility_launches", 0))
    if launches <= 0 or prof.get("visibility_ms", 0) <= 0 or not prof.get("visibility_rays"):
        return None
    avg_ms = prof["visibility_ms"] / launches
    pmc = load_pmc(name)
    if pmc.get("spp") not in (None, spp):
        pmc = {}
    traffic = pmc.get("visibility_hbm_bytes_per_launch")
    roof = {
        "bound": "hbm", "kernel": "k_shadow_walk (visibility)",
        "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic,
        "avg_launch_ms": avg_ms, "launches": launches, "total_ms": prof["visibility_ms"],
        "rays_per_launch": prof["visibility_rays"] / launches,
        "l2_hit_rate": pmc.get("visibility_l2_hit_rate"),
        "profile": "profiles/pmc_%s.json" % name if pmc else None,
    }
    if traffic:
        roof["hbm_side"] = {"bytes_per_launch": traffic, "GB_per_s": traffic / (avg_ms * 1e-3) / 1e9,
                            "frac": traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "how": pmc.get("hbm_bytes_how")}
    if counts is not None and counts.rays_shadow:
        box_bytes = 10.0 if info.get("n_oct_nodes", 0) > 0 else 28.0
        per_ray = (box_bytes * counts.shadow_node_boxes + 36.0 * counts.shadow_tri_tests) / counts.rays_shadow + 29.0
        roof["node_boxes_per_ray"] = counts.shadow_node_boxes / counts.rays_shadow
        roof["tri_tests_per_ray"] = counts.shadow_tri_tests / counts.rays_shadow
        roof["algorithmic_bytes_per_ray"] = per_ray
        roof["achieved"] = per_ray * roof["rays_per_launch"] / (avg_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    valu = pmc.get("visibility_valu_insts_per_launch")
    if valu:
        rate = valu / (avg_ms * 1e-3) / 1e9
        roof["valu_issue"] = {
            "achieved": rate, "unit": "G wave-instr/s",
            "peak_datasheet": VALU_PEAK_DATASHEET_GINST, "frac_datasheet": rate / VALU_PEAK_DATASHEET_GINST,
            "sustained_under_load": VALU_SUSTAINED_GINST,
            "active_lanes_of_64": pmc.get("visibility_active_lanes"),
            "lds_bank_conflict_frac": pmc.get("visibility_lds_conflict_frac"),
            "wave_time_waiting_on_memory": pmc.get("visibility_wait_any_frac"),
        }
    roof["binding_bound"] = "scattered_gather_rate (L1/L2 requests per second) and vector-instruction issue, not HBM bytes"
    if roof.get("frac") is not None:
        roof["note"] = ("the scene is served by L2 and the Infinity Cache as well as HBM, so the algorithmic bytes per "
                        "second may exceed the HBM peak; hbm_side is what crossed the memory side of L2")
    return roof


def run_config(name, cfg, spp_total, scaling, args, steps, warmup, count, rank, local_rank, world, dev):
    """Builds the scene, renders `warmup` untimed and `steps` timed frames of it and returns the
    measurements (on every rank; sums and the slowest rank's time are already reduced)."""
    import torch
    import torch.distributed as dist
    from pathtracer_amd import api, host, distributed

    W, H = cfg["W"], cfg["H"]
    if rank == 0:  # a line per configuration on stderr: a long run shows that it is alive
        print("[bench] %s: building the scene" % name, file=sys.stderr, flush=True)
    if world > 1:
        if scaling == "strong":
            _, spp, _ = distributed.shard(spp_total, rank, world)     # rank r renders indices r, r + world, ...
        else:
            spp = spp_total // world
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

    def step(n_spp):
        accum.zero_()
        c = ds.render_dev(cam, W, H, n_spp, accum.data_ptr(), spp_begin=rank, seed=0, opt=opt, stream=stream)
        if dist.is_initialized():
            dist.all_reduce(accum)
        return c

    # instrumented pass: node boxes and triangles the traversal tests per ray, on the first samples of this
    # same frame (means over tens of millions of rays; they do not depend on how many samples follow)
    counts, count_spp = None, 0
    if count:
        count_spp = min(spp, 4)
        ds.count_stats(True)
        counts = step(count_spp)
        torch.cuda.synchronize()
        ds.count_stats(False)

    for _ in range(warmup):
        step(spp)
    torch.cuda.synchronize()
    if rank == 0:
        print("[bench] %s: %d timed frame(s) of %d spp" % (name, steps, spp), file=sys.stderr, flush=True)
    ds.profile(not args.no_profile)
    ds.read_profile()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step(spp)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ds.read_profile()
    ds.profile(False)

    rays = last.rays_closest + last.rays_shadow
    tot = torch.tensor([dt, float(rays), float(last.paths), 1.0], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    all_rays, all_paths, ranks_seen = float(tot[1]), float(tot[2]), int(round(float(tot[3])))
    K = max(1, steps)
    info = ds.info
    res = {
        "value": all_rays * K / dt / 1e6,
        "ms_per_step": dt / K * 1e3,
        "samples_per_s": all_paths * K / dt,
        "finite": bool(torch.isfinite(accum).all()),
        "ranks_seen": ranks_seen,
        "scaling": scaling,
        "spp_per_gpu": spp, "spp_total": spp * world if scaling == "weak" or world == 1 else spp_total,
        "workload": "%s: %d triangles in %d meshes, %d area light(s), %dx%d, %d spp per GPU%s, %s traversal"
                    % (name, info["n_tris"], info["n_meshes"], info["n_lights"], W, H, spp,
                       " (%d in all, %s scaling)" % (spp_total if scaling == "strong" else spp * world, scaling) if world > 1 else "", mode),
        "rays_per_sample": all_rays / max(1.0, all_paths),
        "bounce_passes": int(last.bounce_passes),
        "max_depth": opt.max_depth, "aperture": float(cam.aperture),
        "kernel_ms_per_step": {"trace_closest": prof["trace_ms"] / K, "shade_nee": prof["shade_ms"] / K,
                               "tail": prof["tail_ms"] / K, "raygen_resolve": prof["other_ms"] / K},
        "roofline": None,
    }
    if not args.no_profile:
        # `roofline` is the block of the kernel the frame spends most of its time in: the closest-hit walk, or
        # for a scene with many lights the visibility walk; the other one rides along
        closest = roofline_block(name, mode, spp, info, prof, counts, count_spp)
        vis = visibility_block(name, spp, info, prof, counts)
        if closest is not None:
            closest["total_ms"] = prof["trace_ms"]
        if vis is not None and closest is not None and vis["total_ms"] > closest["total_ms"]:
            res["roofline"], res["roofline_closest_hit"] = vis, closest
        else:
            res["roofline"] = closest
            if vis is not None:
                res["roofline_visibility"] = vis
    ds.close()
    scene.close()
    return res, (meshes, mats, cam_kw, {k: v for k, v in opt_kw.items() if k not in ("trace_mode", "batch_spp")})


def config_line(name, cfg, r, cpu):
    line = {
        "workload": r["workload"], "value": r["value"], "unit": "Mrays/s", "ms_per_step": r["ms_per_step"],
        "samples_per_s": r["samples_per_s"], "rays_per_sample": r["rays_per_sample"],
        "scaling": r["scaling"], "spp_per_gpu": r["spp_per_gpu"], "named_spp": cfg["named_spp"],
        "max_depth": r["max_depth"], "aperture": r["aperture"],
        "kernel_ms_per_step": r["kernel_ms_per_step"], "finite": r["finite"],
        "roofline": r["roofline"],
    }
    for k in ("roofline_visibility", "roofline_closest_hit"):
        if r.get(k) is not None:
            line[k] = r[k]
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


def claim_stdout():
    """Keeps stdout for the one JSON line: everything else a run writes there (RCCL prints a version banner to
    stdout when a communicator is made) goes to stderr.  Returns the descriptor to hand to emit()."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def emit(fd, obj):
    sys.stdout.flush()
    os.write(fd, (json.dumps(obj) + "\n").encode())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`--gpus N` without a launcher: start the N ranks as fresh child processes (this process has made no GPU
    call), wait for them, relay rank 0's JSON line."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def run_inprocess(args):
    """`--multi inprocess`: one process, pt_render_multi over N devices (one host thread per device, the
    library's own RCCL all-reduce).  The timed region is the whole call, host accumulator included."""
    import numpy as np
    from pathtracer_amd import api, host
    out_fd = claim_stdout()
    n = args.gpus
    if args.share_device:
        os.environ["PT_MULTI_SHARE_DEVICE"] = "1"
    elif api.device_count() < n:
        raise SystemExit("--multi inprocess --gpus %d needs %d devices (found %d); --share-device rehearses on one" % (n, n, api.device_count()))
    name = args.config
    cfg = config_of(name, args)
    spp_total, scaling = samples_for(name, args, n, True)
    meshes, mats = cfg["scene"]()
    scene = host.Scene.from_meshes(meshes, mats)
    if args.share_device:
        copies = [host.Scene.from_meshes(meshes, mats) for _ in range(n)]
        scenes = [c.device_scene(0) for c in copies]
    else:
        scenes = [scene.device_scene(g) for g in range(n)]
    cam_kw = dict(cfg["cam"])
    cam = api.make_camera(**cam_kw)
    opt_kw = dict(cfg.get("opt", {}))
    opt = api.make_options(**opt_kw)
    W, H = cfg["W"], cfg["H"]
    acc = np.zeros((H, W, 3), np.float32)
    for _ in range(args.warmup):
        acc[:] = 0
        api.render_multi(scenes, cam, W, H, spp_total, seed=0, opt=opt, accum=acc)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc[:] = 0
        _, c = api.render_multi(scenes, cam, W, H, spp_total, seed=0, opt=opt, accum=acc)
    dt = time.perf_counter() - t0
    K = max(1, args.steps)
    rays = c.rays_closest + c.rays_shadow
    out = {
        "metric": "Mrays/s at 1080p" if H == 1080 else "Mrays/s", "value": rays * K / dt / 1e6, "unit": "Mrays/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%d, %d spp dealt over %d device(s) by pt_render_multi (one process, one host thread per "
                               "device, RCCL all-reduce%s), host accumulator" % (name, W, H, spp_total, n,
                                                                             " replaced by an on-device sum: all copies share device 0" if args.share_device else ""),
                   "multi": "inprocess", "share_device": bool(args.share_device)},
        "samples_per_s": c.paths * K / dt, "finite": bool(np.isfinite(acc).all()), "ranks_seen": n,
    }
    emit(out_fd, out)


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and args.multi == "inprocess":
        return run_inprocess(args)
    if args.gpus > 1 and world_env is None:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist

    out_fd = claim_stdout()
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher (WORLD_SIZE set) the process group is made even for one rank, so that a one-GPU box
    # runs the same RCCL calls (group of one) as a node does
    if world_env is not None:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    name = args.config
    cfg = config_of(name, args)
    spp_total, scaling = samples_for(name, args, world, True)
    res, (meshes, mats, cam_kw, opt_kw) = run_config(name, cfg, spp_total, scaling, args, args.steps, args.warmup, not args.no_count,
                                                     rank, local_rank, world, dev)
    out = {
        "metric": "Mrays/s at 1080p" if cfg["H"] == 1080 else "Mrays/s",
        "value": res["value"], "unit": "Mrays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": scaling if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": res["workload"], "rays_per_sample": res["rays_per_sample"],
                   "bounce_passes": res["bounce_passes"], "max_depth": res["max_depth"], "aperture": res["aperture"]},
        "samples_per_s": res["samples_per_s"], "finite": res["finite"], "ranks_seen": res["ranks_seen"],
        "kernel_ms_per_step": res["kernel_ms_per_step"],
    }
    if res["roofline"] is not None:
        out["roofline"] = res["roofline"]
    for k in ("roofline_visibility", "roofline_closest_hit"):
        if res.get(k) is not None:
            out[k] = res[k]
    do_cpu = world == 1 and rank == 0 and not args.no_cpu
    if do_cpu:
        out["cpu_baseline"] = cpu_baseline(meshes, mats, cfg, cam_kw, opt_kw)
    del meshes, mats

    # the other named configurations, each measured like the main one: same protocol (warm-up, timed frames
    # between barriers, slowest rank's time), its own roofline block, its own time-boxed CPU baseline
    if args.others and name == "box1k":
        lines = []
        for other in OTHERS:
            c = config_of(other, args)
            tot, sc = samples_for(other, args, world, False)
            r, (m2, mt2, ck, ok) = run_config(other, c, tot, sc, args, args.other_steps, 1, not args.no_count, rank, local_rank, world, dev)
            cpu = cpu_baseline(m2, mt2, c, ck, ok, budget_s=10.0) if do_cpu else None
            del m2, mt2
            lines.append(config_line(other, c, r, cpu))
        out["configs"] = lines
    if rank == 0:
        emit(out_fd, out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
