#!/usr/bin/env python3
"""Times rt_render, the entry point that hands back host buffers (allocation,
kernel, device-to-host copy), next to the kernel alone, on the headline workload."""
import sys
import time

sys.path.insert(0, ".")
from raytracing_amd import Renderer, Scene, make_params
from raytracing_amd.scenes import MAIN_RS_LIGHT_SHAPE, cornell_box_scene

with Scene(0x5EED0001) as sc:
    cam, world = cornell_box_scene(1.0)
    sc.light_xz_rect(*MAIN_RS_LIGHT_SHAPE)
    flat = sc.flatten(world)
    r = Renderer(0)
    r.upload_scene(flat)
    p = make_params(1024, 1024, 1000, seed=0x5EED0001)
    r.render(cam, p)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r.render(cam, p)
        best = min(best, time.perf_counter() - t0)
    print("rt_render 1024x1024@1000 host buffers: %.2f ms per image = %.0f Mpaths/s" % (best * 1e3, 1024 * 1024 * 1000 / best / 1e6))
    r.close()
