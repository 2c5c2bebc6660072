#!/usr/bin/env python3
"""Throughput of the render kernel on the other named configurations of
BASELINE.json (bench.py stays the headline Cornell number).  One JSON line per
scene: Mpaths/s, Mrays/s, kernel ms, event counts.

  python bench_scenes.py --scene random --nx 400 --ny 225 --spp 16
  python bench_scenes.py --scene final --nx 1920 --ny 1080 --spp 64
  python bench_scenes.py --scene stress --spheres 1048576 --nx 3840 --ny 2160 --spp 4
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
        cam, world = S.final_scene(ratio)
        return cam, world, S.FINAL_LIGHT_SHAPE, 0
    if name == "stress":
        cam, world = S.sphere_stress_scene(ratio, args.spheres)
        return cam, world, S.STRESS_LIGHT_SHAPE, 0
    raise SystemExit("unknown scene " + name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--nx", type=int, default=400)
    ap.add_argument("--ny", type=int, default=225)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--spheres", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lbvh", action="store_true", help="rebuild the world tree on the device (rt_build_lbvh): a separate variant")
    ap.add_argument("--cpu-spp", type=int, default=0, help="also time the f64 oracle at this many spp (8 threads)")
    args = ap.parse_args()

    import torch
    torch.cuda.set_device(0)  # torch first: it brings up the HIP runtime the library then shares
    from raytracing_amd import Renderer, Scene, make_params
    t0 = time.perf_counter()
    sc = Scene(0x5EED0000)
    with sc:
        cam, world, light, sky = build(args.scene, float(args.nx) / args.ny, args)
        if light is not None:
            sc.light_xz_rect(*light)
        flat = sc.flatten(world)
    t_build = time.perf_counter() - t0
    r = Renderer(0)
    t_lbvh = None
    if args.lbvh:
        r.build_lbvh(flat)  # first call pays for library initialisation
        t1 = time.perf_counter()
        flat = r.build_lbvh(flat)
        t_lbvh = time.perf_counter() - t1
    r.upload_scene(flat)
    p = make_params(args.nx, args.ny, args.spp, sky=sky)
    best = None
    for _ in range(args.reps):
        _, stats = r.render(cam, p, want_stats=True)
        if best is None or stats.kernel_ms < best.kernel_ms:
            best = stats
    paths = args.nx * args.ny * args.spp
    # the production build of the kernel (no event counters), device buffers, HIP events on the launch stream
    dev = torch.device("cuda", 0)
    buf = torch.empty(args.ny * args.nx * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    prod_ms = None
    for _ in range(args.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        prod_ms = ms if prod_ms is None else min(prod_ms, ms)
    line = {
        "scene": args.scene, "nx": args.nx, "ny": args.ny, "spp": args.spp, "nodes": flat.n_nodes, "prims": flat.n_prims,
        "tree": "device LBVH (variant)" if args.lbvh else "reference median split", "lbvh_build_s": None if t_lbvh is None else round(t_lbvh, 4),
        "build_s": round(t_build, 2), "kernel_ms": round(prod_ms, 3), "counting_kernel_ms": round(best.kernel_ms, 3),
        "Mpaths_per_s": round(paths / prod_ms / 1e3, 2), "Mrays_per_s": round(best.rays / prod_ms / 1e3, 2),
        "rays_per_path": round(best.rays / float(paths), 3), "node_tests_per_ray": round(best.node_tests / float(best.rays), 2),
        "note": "kernel_ms times the production build; the event counts come from a separate launch of the counting build",
    }
    if args.cpu_spp > 0:
        from oracle import binding as orc
        cp = make_params(args.nx, args.ny, args.cpu_spp, sky=sky)
        c0 = time.perf_counter()
        orc.render(flat, cam.c, cp, mode=orc.ORC_F64, n_threads=8, split_samples=True, want_sumsq=False)
        dt = time.perf_counter() - c0
        line["cpu_oracle_Mpaths_per_s_8_threads"] = round(args.nx * args.ny * args.cpu_spp / dt / 1e6, 3)
    print(json.dumps(line))
    r.close()


if __name__ == "__main__":
    main()
