#!/usr/bin/env python3
"""Throughput of the render kernel on the other named configurations of
BASELINE.json (bench.py stays the headline Cornell number).  One JSON line per
scene: Mpaths/s, Mrays/s, kernel ms, event counts.

  python bench_scenes.py --scene random --nx 400 --ny 225 --spp 16
  python bench_scenes.py --scene final --nx 1920 --ny 1080 --spp 64
  python bench_scenes.py --scene stress --spheres 1048576 --nx 3840 --ny 2160 --spp 4

Several GPUs of one node (one process per GPU, as bench.py): pixel tiles or
sample ranges per rank, one RCCL reduce(sum) of the fp32 buffers per image; the
image and sample count stay fixed (strong scaling) and the line reports the job:

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bench_scenes.py \
      --gpus 8 --shard tiles --scene final --nx 1920 --ny 1080 --spp 4096
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(name, ratio, args):
    from raytracing_amd import scenes as S
    if name == "cornell":
        cam, world = S.cornell_box_scene(ratio)
        return cam, world, S.MAIN_RS_LIGHT_SHAPE, 0
    if name == "random":
        cam, world = S.random_scene(ratio)
        return cam, world, None, 0
    if name == "smoke":
        cam, world = S.cornell_smoke_scene(ratio, smoke=True)
        return cam, world, S.SMOKE_LIGHT_SHAPE, 0
    if name == "final":
        # RT_FINAL_VARIANT=nomarble,nomedia: timing experiments on what those parts cost
        v = os.environ.get("RT_FINAL_VARIANT", "")
        cam, world = S.final_scene(ratio, marble="nomarble" not in v, media="nomedia" not in v)
        return cam, world, S.FINAL_LIGHT_SHAPE, 0
    if name == "stress":
        cam, world = S.sphere_stress_scene(ratio, args.spheres)
        return cam, world, S.STRESS_LIGHT_SHAPE, 0
    raise SystemExit("unknown scene " + name)


# The named configurations of BASELINE.json other than the headline (bench.py's), as bench.py's `other_configs` runs them:
# the configuration's own scene and frame, a sample count that keeps the default bench run short (throughput in paths/s
# does not depend on it beyond the end-of-launch tail), and the bounded sample the CPU restatement is timed on.
OTHER_CONFIGS = [
    {"name": "C1 Book-1 random spheres 400x225 @ 16 spp", "scene": "random", "nx": 400, "ny": 225, "spp": 16,
     "cpu": (400, 225, 256), "named_spp": 16},
    {"name": "C3 Book-2 final scene 1920x1080 (named at 4096 spp)", "scene": "final", "nx": 1920, "ny": 1080, "spp": 256,
     "cpu": (1920, 1080, 8), "named_spp": 4096},
    {"name": "C4 2^20-sphere BVH stress 3840x2160 @ 64 spp", "scene": "stress", "nx": 3840, "ny": 2160, "spp": 64,
     "cpu": (960, 540, 1), "named_spp": 64},
    {"name": "C5 Cornell + glass sphere + smoke 7680x4320 (named at 16384 spp)", "scene": "smoke", "nx": 7680, "ny": 4320, "spp": 64,
     "cpu": (1920, 1080, 64), "named_spp": 16384},
]


def measure_config(cfg, device_index=0, reps=3, cpu=True, spheres=1 << 20, seed=0x5EED0000):
    """One entry of bench.py's `other_configs`: the production kernel timed with HIP events on its launch stream, the
    event counts of the same launch from the counting build, the roofline on those counts, and the CPU restatement
    (f64, the reference's 8-thread sample split) on a bounded sample of the same scene."""
    import statistics
    import torch
    from raytracing_amd import Renderer, Scene, make_params
    from bench import VALU_PEAK_TFLOPS, HBM_PEAK_GBS, algorithmic_flops

    class _A(object):
        pass
    a = _A()
    a.spheres = spheres
    nx, ny, spp = cfg["nx"], cfg["ny"], cfg["spp"]
    t0 = time.perf_counter()
    sc = Scene(seed)
    with sc:
        cam, world, light, sky = build(cfg["scene"], float(nx) / ny, a)
        if light is not None:
            sc.light_xz_rect(*light)
        flat = sc.flatten(world)
    t_build = time.perf_counter() - t0
    r = Renderer(device_index)
    r.upload_scene(flat)
    upload_ms = r.upload_ms
    p = make_params(nx, ny, spp, sky=sky)
    dev = torch.device("cuda", device_index)
    buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    times = []
    for k in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        if k > 0:
            times.append(e0.elapsed_time(e1))
    ms = statistics.median(times)
    checksum = float(buf.double().sum().item())
    del buf
    _, st = r.render(cam, p, want_stats=True)
    r.close()
    paths = float(nx) * ny * spp
    flops = float(algorithmic_flops(st))
    achieved = flops / (ms * 1e-3) / 1e12
    out = {
        "config": cfg["name"], "scene": cfg["scene"], "nx": nx, "ny": ny, "spp": spp, "named_spp": cfg["named_spp"],
        "nodes": flat.n_nodes, "prims": flat.n_prims, "scene_build_s": round(t_build, 2),
        "upload_ms": {"host_build": round(upload_ms[0], 2), "device_copies": round(upload_ms[1], 2),
                      "what": "rt_upload_scene: the device's own tree views built on the host (SAH + four-wide interior over the given "
                              "leaves, scan array, per-primitive records), then allocation and copies"},
        "kernel_ms": round(ms, 3), "Mpaths_per_s": round(paths / ms / 1e3, 2), "Mrays_per_s": round(st.rays / ms / 1e3, 2),
        "sec_to_image_at_named_spp": round(ms * 1e-3 * cfg["named_spp"] / spp, 4),
        "rays_per_path": round(st.rays / paths, 3), "node_tests_per_ray": round(st.node_tests / float(st.rays), 2),
        "image_checksum": checksum,
        "roofline": {"bound": "valu", "kernel": "render_kernel", "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / VALU_PEAK_TFLOPS, 4), "flops_per_path": round(flops / paths, 1),
                     "definition": "event counts of this launch (rt_stats) x the per-event constants of bench.py / kernel time"},
    }
    if cfg["scene"] == "stress":
        # The traversal of this scene is the memory-side configuration: its fabric traffic comes from the committed PMC
        # profile of this exact launch (bytes per launch do not depend on the run), divided by this run's kernel time.
        tf = os.path.join(ROOT, "profiles", "r3", "stress_traffic.json")
        mem = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
               "request_bytes_per_launch": int(32 * st.node_tests + 20 * st.sphere_tests),
               "note": "achieved = fabric read bytes per launch (FETCH_SIZE x 2, PMC profile of the same launch) / kernel time; "
                       "request bytes (32 B per box test + 20 B per sphere test) are nearly all served by L1/L2 and are not traffic"}
        if os.path.exists(tf):
            t = json.load(open(tf))
            if (t.get("nx"), t.get("ny"), t.get("spp")) == (nx, ny, spp):
                mem["traffic"] = int(t["read_bytes"] + t.get("write_bytes", 0))
                mem["achieved"] = round(mem["traffic"] / (ms * 1e-3) / 1e9, 1)
                mem["frac"] = round(mem["achieved"] / HBM_PEAK_GBS, 4)
        out["roofline_memory"] = mem
    if cpu:
        from oracle import binding as orc
        cnx, cny, cspp = cfg["cpu"]
        if cnx * ny == cny * nx:  # same aspect ratio: the same camera and scene
            ccam, cflat, csky = cam, flat, sky
        else:
            sc2 = Scene(seed)
            with sc2:
                ccam, cworld, clight, csky = build(cfg["scene"], float(cnx) / cny, a)
                if clight is not None:
                    sc2.light_xz_rect(*clight)
                cflat = sc2.flatten(cworld)
        cp = make_params(cnx, cny, cspp, sky=csky)
        c0 = time.perf_counter()
        orc.render(cflat, ccam.c, cp, mode=orc.ORC_F64, n_threads=8, split_samples=cspp >= 8, want_sumsq=False)
        dt = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": round(cnx * cny * cspp / dt / 1e6, 4), "unit": "Mpaths/s", "cores": 8, "kind": "port",
                               "sample": "same scene at %dx%d, %d spp (%.1f s); f64 restatement of the reference algorithm on 8 threads "
                                         "(main.rs:70), not the Rust binary" % (cnx, cny, cspp, dt)}
    return out


# The two named configurations that are defined on several GPUs (BASELINE.json configs 3 and 5), as bench.py runs them
# when it is launched with more than one rank: the final scene by pixel tiles with a gather of the tiles (its named
# 4096 spp), the 8K smoke frame by sample ranges with a sum of the frames (1024 spp of its named 16384: paths/s does not
# depend on the count, and the one-GPU time the efficiency is taken against stays under two seconds).
MULTI_CONFIGS = [
    {"name": "C3 Book-2 final scene 1920x1080 @ 4096 spp: pixel tiles per GPU, tiles gathered on rank 0", "scene": "final",
     "nx": 1920, "ny": 1080, "spp": 4096, "named_spp": 4096, "shard": "tiles"},
    {"name": "C5 Cornell + glass sphere + smoke 7680x4320 (named at 16384 spp): sample range per GPU, frames summed on rank 0",
     "scene": "smoke", "nx": 7680, "ny": 4320, "spp": 1024, "named_spp": 16384, "shard": "samples"},
]


def measure_config_multi(cfg, rank, n_ranks, device_index, rehearse=False, reps=2, spp=None, seed=0x5EED0000):
    """One named configuration on all ranks of the job (every rank calls this; only rank 0 gets the result, the others
    None).  Per image: every rank renders its shard into its own frame (kernel time by HIP events on the launch
    stream), then the exchange (TileGather for pixel tiles, one reduce(sum) for sample ranges); the job's time is the
    slowest rank's wall clock between two barriers.  Rank 0 then renders the whole image alone, which gives the one-GPU
    time the efficiency is taken against in the same run and the image the exchanged frame is checked against."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import Renderer, Scene, make_params
    from raytracing_amd.distributed import TileGather, reduce_to_root, shard_params

    class _A(object):
        pass
    a = _A()
    a.spheres = 1 << 20
    nx, ny = cfg["nx"], cfg["ny"]
    spp = int(spp or cfg["spp"])
    shard = cfg["shard"]
    sc = Scene(seed)
    with sc:
        cam, world, light, sky = build(cfg["scene"], float(nx) / ny, a)
        if light is not None:
            sc.light_xz_rect(*light)
        flat = sc.flatten(world)
    r = Renderer(device_index)
    r.upload_scene(flat)
    whole_p = make_params(nx, ny, spp, sky=sky)
    p = shard_params(whole_p, rank, n_ranks, shard)
    dev = torch.device("cuda", device_index)
    buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    gather = None
    if shard == "tiles":
        gather = TileGather(nx, ny) if rehearse else TileGather(nx, ny, device=dev)

    def fence():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    def exchange():
        if rehearse:  # gloo moves host tensors
            host = buf.cpu()
            (gather.gather if shard == "tiles" else reduce_to_root)(host)
            buf.copy_(host)
        elif shard == "tiles":
            gather.gather(buf)
        else:
            reduce_to_root(buf)

    best = None
    for k in range(reps + 1):
        fence()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        exchange()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        fence()
        t3 = time.perf_counter()
        mine = torch.tensor([e0.elapsed_time(e1), (t2 - t1) * 1e3, (t3 - t0) * 1e3], dtype=torch.float64)
        every = [torch.zeros(3, dtype=torch.float64) for _ in range(n_ranks)]
        if rehearse:
            dist.all_gather(every, mine)
        else:
            every_d = [t.to(dev) for t in every]
            dist.all_gather(every_d, mine.to(dev))
            every = [t.cpu() for t in every_d]
        job_ms = max(float(t[2]) for t in every)
        if k > 0 and (best is None or job_ms < best["ms_per_image"]):
            best = {"ms_per_image": job_ms, "kernel_ms_per_rank": [round(float(t[0]), 3) for t in every],
                    "exchange_ms_per_rank": [round(float(t[1]), 3) for t in every]}
    # the whole image on rank 0 alone: the one-GPU time of this run, and the image the exchanged frame must equal
    one_gpu_ms = None
    check = None
    if rank == 0:
        whole = torch.empty_like(buf)
        times = []
        for k in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            r.render_device(cam, whole_p, whole.data_ptr(), None, stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            times.append(e0.elapsed_time(e1))
        one_gpu_ms = min(times)
        diff = (buf - whole).abs()
        tol = 1e-5 * whole.abs() + 1e-5
        check = {"values_beyond_fp32_order": int((diff > tol).sum().item()), "max_abs_diff": float(diff.max().item()),
                 "values": int(whole.numel()), "sum": float(whole.double().sum().item())}
        del whole, diff, tol
    fence()
    r.close()
    if rank != 0:
        return None
    paths = float(nx) * ny * spp
    best["ms_per_image"] = round(best["ms_per_image"], 3)
    out = {
        "config": cfg["name"], "scene": cfg["scene"], "nx": nx, "ny": ny, "spp": spp, "named_spp": cfg["named_spp"], "n_gpus": n_ranks,
        "sharding": ("REHEARSAL: ranks share one GPU, gloo exchange of host tensors; " if rehearse else "") + shard,
        "scaling": "strong",
    }
    out.update(best)
    out["Mpaths_per_s"] = round(paths / best["ms_per_image"] / 1e3, 2)
    out["sec_to_image_at_named_spp"] = round(best["ms_per_image"] * 1e-3 * cfg["named_spp"] / spp, 4)
    out["one_gpu_ms"] = round(one_gpu_ms, 3)
    out["efficiency_vs_1gpu"] = round(one_gpu_ms / (n_ranks * best["ms_per_image"]), 4)
    out["exchange_bytes_per_rank"] = gather.bytes_sent_per_rank() if shard == "tiles" else nx * ny * 12
    out["image_vs_one_gpu"] = check
    out["note"] = ("ms_per_image: slowest rank, wall clock between barriers, kernel + exchange; one_gpu_ms: the whole image on rank 0 "
                   "alone in the same run (HIP events); efficiency = one_gpu_ms / (n_gpus x ms_per_image)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--nx", type=int, default=400)
    ap.add_argument("--ny", type=int, default=225)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--spheres", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lbvh", action="store_true", help="rebuild the world tree on the device (rt_build_lbvh): a separate variant")
    ap.add_argument("--keep-tree", action="store_true",
                    help="RT_OPT_KEEP_TREE: the device walks the interior nodes as given (the reference's tree, or with --lbvh the device-built one) "
                         "collapsed four-wide, instead of the surface-area tree rt_upload_scene builds by default")
    ap.add_argument("--cpu-spp", type=int, default=0, help="also time the f64 oracle at this many spp (8 threads)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--shard", choices=("tiles", "samples"), default="tiles", help="how ranks split the image when --gpus > 1")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="ranks share GPU 0 and exchange host tensors over gloo: checks the multi-rank flow, not a benchmark")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    n_ranks = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if n_ranks != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d (launch with torch.distributed.run)" % (n_ranks, args.gpus))
    torch.cuda.set_device(local_rank)  # torch first: it brings up the HIP runtime the library then shares
    if n_ranks > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from raytracing_amd import Renderer, Scene, make_params
    from raytracing_amd.distributed import TileGather, reduce_to_root, shard_params
    from bench import VALU_PEAK_TFLOPS, HBM_PEAK_GBS, algorithmic_flops
    t0 = time.perf_counter()
    sc = Scene(0x5EED0000)
    with sc:
        cam, world, light, sky = build(args.scene, float(args.nx) / args.ny, args)
        if light is not None:
            sc.light_xz_rect(*light)
        flat = sc.flatten(world)
    t_build = time.perf_counter() - t0
    r = Renderer(local_rank)
    t_lbvh = None
    if args.lbvh:
        r.build_lbvh(flat)  # first call pays for library initialisation
        t1 = time.perf_counter()
        flat = r.build_lbvh(flat)
        t_lbvh = time.perf_counter() - t1
    if args.keep_tree:
        from raytracing_amd._lib import RT_OPT_KEEP_TREE
        r.set_option(RT_OPT_KEEP_TREE, 1)
    r.upload_scene(flat)
    r.upload_scene(flat)  # (the second upload's times: the first pays for the allocator's first pages)
    upload_ms = r.upload_ms
    p = shard_params(make_params(args.nx, args.ny, args.spp, sky=sky), rank, n_ranks, args.shard)
    best = None
    for _ in range(args.reps):
        _, stats = r.render(cam, p, want_stats=True)
        if best is None or stats.kernel_ms < best.kernel_ms:
            best = stats
    paths = args.nx * args.ny * args.spp  # of the whole job
    # the production build of the kernel (no event counters), device buffers, HIP events on the launch stream
    dev = torch.device("cuda", local_rank)
    buf = torch.empty(args.ny * args.nx * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def fence():
        torch.cuda.synchronize(dev)
        if n_ranks > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gather_dev = gather_host = None
    if n_ranks > 1 and args.shard == "tiles":
        if args.rehearse_on_one_gpu:
            gather_host = TileGather(args.nx, args.ny)
        else:
            gather_dev = TileGather(args.nx, args.ny, device=dev)
    prod_ms = None   # this rank's kernel, best of the repetitions
    job_ms = None    # kernel + exchange, slowest rank, best of the repetitions
    for _ in range(args.reps + 1):
        fence()
        t1 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        if n_ranks > 1:
            # sample ranges are summed; tile shards are gathered (each rank sends its own pixels only)
            if args.rehearse_on_one_gpu:
                host = buf.cpu()
                if args.shard == "tiles":
                    gather_host.gather(host)
                else:
                    reduce_to_root(host)
                buf.copy_(host)
            elif args.shard == "tiles":
                gather_dev.gather(buf)
            else:
                reduce_to_root(buf)
        fence()
        dt = time.perf_counter() - t1
        if n_ranks > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        ms = e0.elapsed_time(e1)
        prod_ms = ms if prod_ms is None else min(prod_ms, ms)
        job_ms = dt * 1e3 if job_ms is None else min(job_ms, dt * 1e3)
    # The rehearsal's image check: the root's frame after the exchange against the whole image rendered by the root alone
    # (tile shards: the same samples of the same pixels, equal to fp32 summation order of the kernel's own accumulation;
    # sample ranges: to the order in which the ranks' sums are added).
    image_check = None
    if n_ranks > 1 and args.rehearse_on_one_gpu and rank == 0:
        whole = torch.empty_like(buf)
        r.render_device(cam, make_params(args.nx, args.ny, args.spp, sky=sky), whole.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        diff = (buf - whole).abs()
        tol = 1e-5 * whole.abs() + 1e-5
        image_check = {"max_abs_diff": float(diff.max().item()), "values_beyond_fp32_order": int((diff > tol).sum().item()),
                       "values": int(whole.numel()), "sum": float(whole.double().sum().item())}
        if image_check["values_beyond_fp32_order"] > 0:
            raise SystemExit("rehearsal: the exchanged frame differs from the one-rank image: %r" % (image_check,))
    # totals over the ranks of the event counts (each rank counted its own share)
    rays, node_tests, flops = float(best.rays), float(best.node_tests), float(algorithmic_flops(best))
    sphere_tests = float(best.sphere_tests)
    if n_ranks > 1:
        t = torch.tensor([rays, node_tests, flops, sphere_tests], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rays, node_tests, flops, sphere_tests = (float(x) for x in t.tolist())
    if rank != 0:
        r.close()
        if n_ranks > 1:
            dist.destroy_process_group()
        return
    t_ms = prod_ms if n_ranks == 1 else job_ms
    # Roofline of what the kernel itself evaluated (its event counters x the per-event constants of bench.py); for the
    # sphere stress scene also the algorithmic bytes of its node and sphere fetches against the HBM peak.
    roofline = {"bound": "valu", "achieved": round(flops / (t_ms * 1e-3) / 1e12, 3), "peak": VALU_PEAK_TFLOPS * n_ranks, "unit": "TFLOP/s"}
    roofline["frac"] = round(roofline["achieved"] / roofline["peak"], 4)
    if args.scene == "stress":
        # request bytes are not traffic (nearly all are served by L1/L2): they are listed, not compared with a peak; the
        # fabric traffic of this scene comes from the PMC profile (tools/profile_stress.sh, profiles/r3/stress_summary.txt)
        roofline["request_bytes_per_launch"] = int(32.0 * node_tests + 20.0 * sphere_tests)
    line = {
        "scene": args.scene, "nx": args.nx, "ny": args.ny, "spp": args.spp, "nodes": flat.n_nodes, "prims": flat.n_prims,
        "tree": "device LBVH (variant)" if args.lbvh else "reference median split", "lbvh_build_s": None if t_lbvh is None else round(t_lbvh, 4),
        "walked": "the given interior, collapsed four-wide (RT_OPT_KEEP_TREE)" if args.keep_tree else "rt_upload_scene's own surface-area tree over the given leaves",
        "upload_ms": {"host_build": round(upload_ms[0], 2), "device_copies": round(upload_ms[1], 2)},
        "build_s": round(t_build, 2), "kernel_ms": round(prod_ms, 3), "counting_kernel_ms": round(best.kernel_ms, 3),
        "n_gpus": n_ranks, "sharding": "single GPU" if n_ranks == 1 else ("REHEARSAL on one GPU, " if args.rehearse_on_one_gpu else "") + args.shard,
        "ms_per_image": round(t_ms, 3),
        "Mpaths_per_s": round(paths / t_ms / 1e3, 2), "Mrays_per_s": round(rays / t_ms / 1e3, 2),
        "rays_per_path": round(rays / float(paths), 3), "node_tests_per_ray": round(node_tests / rays, 2),
        "roofline": roofline,
        "note": "one GPU: the production kernel by HIP events; several: kernel + exchange (sum of sample ranges, gather of tile shards), slowest rank, wall clock between barriers; "
                "kernel_ms is rank 0's kernel; the event counts come from a separate launch of the counting build",
    }
    if image_check is not None:
        line["rehearsal_image_check"] = image_check
    if args.cpu_spp > 0:
        from oracle import binding as orc
        cp = make_params(args.nx, args.ny, args.cpu_spp, sky=sky)
        c0 = time.perf_counter()
        orc.render(flat, cam.c, cp, mode=orc.ORC_F64, n_threads=8, split_samples=True, want_sumsq=False)
        dt = time.perf_counter() - c0
        line["cpu_oracle_Mpaths_per_s_8_threads"] = round(args.nx * args.ny * args.cpu_spp / dt / 1e6, 3)
    print(json.dumps(line))
    r.close()
    if n_ranks > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
