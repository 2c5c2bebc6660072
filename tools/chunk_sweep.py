#!/usr/bin/env python3
"""How the stack kernel's item sizes (RT_WANT_ROUNDS, RT_MIN_CHUNK_SPP, RT_STACK_LAST_SPP) set the time of short launches:
the final scene at 1920x1080, whole image at 512 and 64 samples, and one tile shard of eight at 512.
tools/chunk_sweep.py  (on the GPU box)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import Renderer, Scene, make_params, scenes  # noqa: E402

nx, ny = 1920, 1080
with Scene(0x5EED0000) as sc:
    cam, world = scenes.final_scene(float(nx) / ny)
    sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
    flat = sc.flatten(world)
r = Renderer(0)
r.upload_scene(flat)


def timed(p):
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        r.render(cam, p)
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


cases = (("whole512", make_params(nx, ny, 512, seed=1)), ("whole64", make_params(nx, ny, 64, seed=1)),
         ("tile1of8", make_params(nx, ny, 512, seed=1, shard_index=3, shard_count=8)))
for rounds in ("30", "12", "6", "3"):
    for min_spp in ("16", "32", "64"):
        for last in ("4", "8"):
            os.environ["RT_WANT_ROUNDS"] = rounds
            os.environ["RT_MIN_CHUNK_SPP"] = min_spp
            os.environ["RT_STACK_LAST_SPP"] = last
            print("rounds %s min_spp %s last %s: " % (rounds, min_spp, last) + "  ".join("%s %.1f" % (n, timed(p)) for n, p in cases), flush=True)
