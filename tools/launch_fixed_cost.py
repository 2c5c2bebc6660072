#!/usr/bin/env python3
"""Device time of one launch against its sample count, per scene: the intercept of the line is what a launch loses to its
start and its end whatever its length.  tools/launch_fixed_cost.py [scene ...]  (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.set_device(0)
from raytracing_amd import Renderer, Scene, make_params, scenes  # noqa: E402

nx, ny = 1920, 1080
dev = torch.device("cuda", 0)
buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
for name in (sys.argv[1:] or ["final", "random", "smoke"]):
    with Scene(0x5EED0000) as sc:
        ratio = float(nx) / ny
        if name == "final":
            cam, world = scenes.final_scene(ratio)
            sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        elif name == "final_plain":
            cam, world = scenes.final_scene(ratio, marble=False, media=False)
            sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        elif name == "smoke":
            cam, world = scenes.cornell_smoke_scene(ratio)
            sc.light_xz_rect(*scenes.SMOKE_LIGHT_SHAPE)
        elif name == "random":
            cam, world = scenes.random_scene(ratio)
        else:
            cam, world = scenes.cornell_box_scene(ratio)
            sc.light_xz_rect(*scenes.MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(world)
    r = Renderer(0)
    r.upload_scene(flat)
    row = []
    for spp in (4, 8, 16, 32, 64, 128, 256):
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            r.render_device(cam, make_params(nx, ny, spp, seed=2), buf.data_ptr(), None, stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            best = min(best, e0.elapsed_time(e1))
        row.append((spp, best))
    slope = (row[-1][1] - row[-2][1]) / (row[-1][0] - row[-2][0])
    print(name, "  ".join("%d: %.2f" % x for x in row), " slope %.3f ms/spp  intercepts %s" %
          (slope, " ".join("%.2f" % (t - slope * s) for s, t in row)), flush=True)
    r.close()
