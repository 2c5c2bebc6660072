#!/usr/bin/env python3
"""rt_upload_scene's time on a named scene, with the library's own breakdown (RT_VERBOSE): tools/upload_time.py stress"""
import os
import sys
import time

if "quiet" not in sys.argv:
    os.environ["RT_VERBOSE"] = "1"  # the library's own breakdown (and, for large trees, its check of the ranks made by threads)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import Renderer, Scene  # noqa: E402
from bench_scenes import build  # noqa: E402


class A(object):
    spheres = 1 << 20


name = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "quiet" else "stress"
t0 = time.perf_counter()
with Scene(0x5EED0000) as sc:
    cam, world, light, sky = build(name, 16.0 / 9.0, A)
    if light is not None:
        sc.light_xz_rect(*light)
    flat = sc.flatten(world)
print("scene construction and flattening (host classes): %.1f ms, %d nodes, %d primitives" % ((time.perf_counter() - t0) * 1e3, flat.n_nodes, flat.n_prims))
r = Renderer(0)
for k in range(5):
    r.upload_scene(flat)
    print("rt_upload_scene: host build %.1f ms, device copies %.1f ms" % r.upload_ms)
r.close()
