#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

One "step" renders the whole Cornell box image of BASELINE.json's configs[1]:
1024x1024 pixels at 1000 samples per pixel (1.05e9 paths), scene and output
buffers resident in HBM.  With N > 1 ranks (one process per GPU, launched by
torch.distributed.run) the samples of every pixel are split into N equal ranges
(what the reference does with its 8 threads, src/main.rs:74-78) and the fp32
accumulation buffers are summed onto rank 0 with one RCCL reduce per step
(src/main.rs:128-133); the total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic fp32 operations per event (add/sub/mul/div/sqrt/compare/min/max/select = 1, FMA = 2,
# sin/cos/rcp = 1, integer and Philox work = 0), derived from the reference source:
FLOPS = {
    "node_tests": 27,      # Aabb::hit, bvh.rs:46-61: 9 per axis
    "rays": 3,             # the three reciprocals of the direction, once per ray
    "rect_tests": 12,      # XY/XZ/YZRect::hit reject path, hitable.rs:163-173
    "sphere_tests": 24,    # Sphere::hit reject path, hitable.rs:50-61
    "instance_ops": 15,    # Translate (3) + RotateY (12) of the ray, hitable.rs:440, 510-517
    "hits": 25,            # p, normal, rotate/translate back, hitable.rs:175-182, 521-528
    "shade_events": 150,   # ONB, mixture sample, both pdf values incl. the light rect test, weight
    "paths": 3,            # fp32 RGB accumulate, main.rs:108
}
VALU_PEAK_TFLOPS = 157.3   # MI355X fp32 vector peak
# Work of the reference algorithm per path on the Cornell box at 1024x1024: event counts of the CPU restatement
# (36.5 Aabb::hit, 8.2 rect tests, 0.8 instance moves, 2.54 hits, 1.81 scattering vertices, 2.81 rays) x FLOPS.
REF_FLOPS_PER_PATH = 1442.3
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(stats):
    return sum(FLOPS[k] * getattr(stats, k) for k in FLOPS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--cpu-spp", type=int, default=256, help="samples per pixel of the CPU baseline's bounded run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the runs of the other named configurations (one GPU: short runs of all four, about a minute; "
                         "several: the final scene by tiles and the 8K smoke frame by sample ranges, a few seconds each)")
    ap.add_argument("--multi-spp", type=int, default=0,
                    help="several ranks: samples per pixel of the other configurations' multi-rank runs (default: 4096 for the final "
                         "scene, 1024 for the 8K smoke frame)")
    ap.add_argument("--no-overlap", action="store_true", help="several ranks: wait for each image's reduce before rendering the next")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks share GPU 0 and exchange over gloo: checks the multi-rank control flow on a one-GPU box, not a benchmark")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from raytracing_amd import Renderer, Scene, make_params
    from raytracing_amd.scenes import MAIN_RS_LIGHT_SHAPE, cornell_box_scene

    nx, ny, spp = args.nx, args.ny, args.spp
    seed = 0x5EED0001
    sc = Scene(seed)
    with sc:
        cam, world_list = cornell_box_scene(float(nx) / float(ny))
        sc.light_xz_rect(*MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(world_list)
    r = Renderer(local_rank)
    r.upload_scene(flat)

    # sample-range split: rank g renders samples [g*spp/N, (g+1)*spp/N) of every pixel
    from raytracing_amd.distributed import shard_params
    params = shard_params(make_params(nx, ny, spp, seed=seed), rank, world, "samples")
    # With several ranks, images alternate between two buffers and the reduce of one image runs (on RCCL's stream) while
    # the next image renders; a buffer is rendered into again only after its reduce has been waited for.
    overlap = world > 1 and not args.no_overlap
    bufs = [torch.empty(ny * nx * 3, dtype=torch.float32, device=dev) for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(bufs)
    stream = torch.cuda.current_stream(dev)
    state = {"images": 0}

    def finish(slot):
        if pending[slot] is not None:
            work, t = pending[slot]
            work.wait()  # RCCL: the current stream waits; gloo: the host does
            if t is not bufs[slot]:
                bufs[slot].copy_(t)
            pending[slot] = None

    def step(ev0=None, ev1=None):
        slot = state["images"] % len(bufs)
        state["images"] += 1
        buf = bufs[slot]
        finish(slot)
        if ev0 is not None:
            ev0.record(stream)
        r.render_device(cam, params, buf.data_ptr(), None, stream.cuda_stream)
        if ev1 is not None:
            ev1.record(stream)
        if world > 1:
            # gloo (the rehearsal) exchanges host tensors; RCCL reduces the device buffer in place
            t = buf.cpu() if args.rehearse_on_one_gpu else buf
            pending[slot] = (dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, async_op=True), t)
            if not overlap:
                finish(slot)

    def fence():
        for slot in range(len(bufs)):
            finish(slot)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def note(what):
        if args.rehearse_on_one_gpu:
            print("[rank %d] %s" % (rank, what), file=sys.stderr, flush=True)

    note("scene uploaded")
    for _ in range(args.warmup):
        step()
    note("warm-up steps done")
    fence()
    note("first fence passed")
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(*events[k])
    fence()
    elapsed = time.perf_counter() - t0
    note("timed steps done")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)
    buf = bufs[(state["images"] - 1) % len(bufs)]  # the last image
    image_checksum = float(buf.double().sum().item()) if rank == 0 else 0.0

    # HBM bytes per launch of the dominant kernel from a committed PMC profile of this exact workload
    # (rocprofv3 FETCH_SIZE / WRITE_SIZE passes; tools/profile.sh), null for any other workload
    traffic = None
    traffic_file = os.path.join(ROOT, "profiles", "r3", "traffic.json")
    if (nx, ny, spp, world) == (1024, 1024, 1000, 1) and os.path.exists(traffic_file):
        for name, t in json.load(open(traffic_file)).items():
            if name.startswith("render_kernel"):
                traffic = int(t["read_bytes"] + t["write_bytes"])

    # event counts of this rank's launch, from the counting build of the same kernel (untimed)
    counts = None
    executed = None
    if not args.no_stats:
        _, stats = r.render(cam, params, want_stats=True)
        counts = stats.as_dict()
        ex_flops = algorithmic_flops(stats)
        executed = {"flops_per_path": round(ex_flops / float(stats.paths), 1), "rays_per_path": round(stats.rays / float(stats.paths), 3),
                    "TFLOP/s": round(ex_flops / (kernel_ms * 1e-3) / 1e12, 3),
                    }

    # CPU baseline (rank 0, single GPU only): the f64 restatement run the reference's way on a bounded sample; its event
    # counters give the reference algorithm's work per path
    cpu = None
    ref_flops_per_path = REF_FLOPS_PER_PATH if (nx, ny) == (1024, 1024) else None
    ref_source = "constant measured with the CPU restatement at 1024x1024, 16 spp"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as orc
        cpu_threads = 8  # the reference's hard-coded thread count, main.rs:70
        cp = make_params(nx, ny, args.cpu_spp, seed=seed)
        orc.lib()
        c0 = time.perf_counter()
        out = orc.render(flat, cam.c, cp, mode=orc.ORC_F64, n_threads=cpu_threads, split_samples=True, want_sumsq=False, want_stats=True)
        cdt = time.perf_counter() - c0
        ref_flops_per_path = algorithmic_flops(out[-1]) / float(out[-1].paths)
        ref_source = "counted by the CPU restatement in this run (%d spp sample)" % args.cpu_spp
        cpu = {
            "value": round(nx * ny * args.cpu_spp / cdt / 1e6, 3), "unit": "Mpaths/s", "cores": cpu_threads, "kind": "port",
            "sample": "same scene and image size at %d spp (%.1f s); f64 restatement of the reference algorithm run the "
                      "reference's way (8 threads, each the whole image at spp/8, then summed), not the Rust binary"
                      % (args.cpu_spp, cdt),
        }
        # the same sample on this job's whole CPU share (16 hardware threads per GPU on the benchmark hosts)
        share = min(16, os.cpu_count() or 1)
        if share > cpu_threads:
            c0 = time.perf_counter()
            orc.render(flat, cam.c, cp, mode=orc.ORC_F64, n_threads=share, split_samples=True, want_sumsq=False, want_stats=False)
            cdt = time.perf_counter() - c0
            cpu["all_cores"] = {"value": round(nx * ny * args.cpu_spp / cdt / 1e6, 3), "unit": "Mpaths/s", "cores": share,
                                "sample": "the same sample on %d threads (%.1f s)" % (share, cdt)}

    # Roofline of the dominant kernel: the algorithmic fp32 operations of what THIS launch evaluated (its event
    # counters x the per-event constants above) over the kernel's duration by HIP events on its launch stream.
    roofline = None
    paths_per_launch = float(nx) * ny * params.spp
    if executed is not None:
        achieved = executed["TFLOP/s"]
        roofline = {
            "bound": "valu", "kernel": "render_kernel", "achieved": round(achieved, 3), "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / VALU_PEAK_TFLOPS, 4), "traffic": traffic,
            "kernel_ms": round(kernel_ms, 3), "flops_per_path": executed["flops_per_path"], "paths_per_launch": int(paths_per_launch),
            "definition": "algorithmic fp32 operations of the events this launch evaluated (rt_stats of the counting build of the "
                          "same kernel: box tests, rect tests, instance moves, hits, scattering vertices, rays, paths, each x its "
                          "per-event constant) / kernel time (HIP events on the launch stream)",
            "rays_per_path": executed["rays_per_path"],
            "algorithmic_hbm_bytes_per_launch": nx * ny * 12,
            "traffic_note": "HBM bytes per launch from the PMC profile of this workload (profiles/r3/traffic.json); the only "
                            "algorithmic HBM stream is the nx*ny*12-byte image",
            "note": "no MFMA and no HBM stream on this path: the scene is 2 KB and lives in SGPRs/LDS; "
                    "the bound is the fp32 vector ALU",
        }
        if ref_flops_per_path is not None:
            ref_t = ref_flops_per_path * paths_per_launch / (kernel_ms * 1e-3) / 1e12
            roofline["reference_work"] = {
                "flops_per_path": round(ref_flops_per_path, 1), "TFLOP/s": round(ref_t, 3), "frac": round(ref_t / VALU_PEAK_TFLOPS, 4),
                "source": ref_source,
                "note": "secondary figure: the work the reference's own algorithm would do for these paths (unpruned BvhNode::hit "
                        "with every Aabb::hit it makes) at this kernel's speed; the kernel does not execute that work"}

    # main.rs:67-155 end to end on the device: render, tonemap, and the copy of the 8-bit image to the host
    end_to_end = None
    if rank == 0 and world == 1:
        u8 = torch.empty(ny * nx * 3, dtype=torch.uint8, device=dev)
        host = torch.empty(ny * nx * 3, dtype=torch.uint8).pin_memory()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            c0 = time.perf_counter()
            r.render_device(cam, params, bufs[0].data_ptr(), None, stream.cuda_stream)
            r.tonemap_device(bufs[0].data_ptr(), nx, ny, spp, u8.data_ptr(), stream.cuda_stream)
            host.copy_(u8, non_blocking=True)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - c0)
        end_to_end = {"seconds": round(sorted(ts)[1], 5), "what": "render + tonemap on the device + copy of the %d-byte 8-bit image "
                      "to pinned host memory (main.rs:67-155), wall clock, median of 3" % (nx * ny * 3)}

    other = None
    if rank == 0 and world == 1 and not args.no_other_configs:
        from bench_scenes import OTHER_CONFIGS, measure_config
        other = []
        for cfg in OTHER_CONFIGS:
            try:
                other.append(measure_config(cfg, device_index=local_rank, cpu=not args.no_cpu_baseline))
            except Exception as e:  # a failed side configuration must not lose the headline line
                other.append({"config": cfg["name"], "error": "%s: %s" % (type(e).__name__, e)})

    # Several ranks: the two named configurations that are defined on several GPUs (final scene by pixel tiles and a
    # gather, the 8K smoke frame by sample ranges and a sum), each with its per-rank kernel time, its exchange time and
    # its efficiency against the whole image rendered by rank 0 alone in the same run.  Every rank takes part.
    if world > 1 and not args.no_other_configs:
        from bench_scenes import MULTI_CONFIGS, measure_config_multi
        other = []
        for cfg in MULTI_CONFIGS:
            try:
                res = measure_config_multi(cfg, rank, world, local_rank, rehearse=args.rehearse_on_one_gpu,
                                           spp=args.multi_spp if args.multi_spp > 0 else None)
            except Exception as e:  # (a rank that fails alone leaves the others in a collective: the job's timeout ends that)
                res = {"config": cfg["name"], "error": "%s: %s" % (type(e).__name__, e)}
            other.append(res)

    if rank == 0:
        paths = float(nx) * ny * spp * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "Mpaths/s, Cornell box %dx%d @ %d spp (sec-to-image = ms_per_step/1000)" % (nx, ny, spp),
            "value": round(paths / elapsed / 1e6, 2), "unit": "Mpaths/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "sec_to_image": round(ms_per_step / 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cornell_box_scene (scenes.rs:276-360) %dx%d @ %d spp, MixturePDF light sampling, "
                                   "max depth 50" % (nx, ny, spp),
                       "sharding": ("REHEARSAL: ranks share one GPU, gloo exchange" if args.rehearse_on_one_gpu else
                                    "sample range per GPU + RCCL reduce(sum) to rank 0" + (", overlapped with the next image's render" if overlap else "")) if world > 1 else "single GPU",
                       "seed": seed},
            "sec_to_image_end_to_end": end_to_end,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "other_configs": other,
            "event_counts_rank0": counts,
            "image_checksum": image_checksum,
        }
        if executed is not None:
            line["Mrays_per_s"] = round(executed["rays_per_path"] * paths / elapsed / 1e6, 2)
        print(json.dumps(line))
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
