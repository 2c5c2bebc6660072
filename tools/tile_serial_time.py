#!/usr/bin/env python3
"""Time of ONE tile's work item alone on the chip (a tile shard of a split so fine that a shard is one tile), over a sample
of the final scene's tiles: the serial chain of a work item, which no amount of parallel hardware shortens.
tools/tile_serial_time.py [spp]  (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.set_device(0)
from raytracing_amd import Renderer, Scene, make_params, scenes  # noqa: E402
from raytracing_amd.distributed import shard_width  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nx, ny = 1920, 1080
dev = torch.device("cuda", 0)
buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
with Scene(0x5EED0000) as sc:
    cam, world = scenes.final_scene(float(nx) / ny)
    sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
    flat = sc.flatten(world)
r = Renderer(0)
r.upload_scene(flat)
count = 240 * 135 + 7  # more shards than tiles, no common factor with 240
assert shard_width(nx, count) == 240
times = []
for t in range(0, 240 * 135, 53):
    best = 1e30
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, make_params(nx, ny, spp, seed=2, shard_index=t, shard_count=count), buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1))
    times.append((best, t % 240, t // 240))
a = np.array([x[0] for x in times])
print("spp %d: %d tiles  median %.3f ms  mean %.3f  p90 %.3f  p99 %.3f  max %.3f" % (spp, len(a), np.median(a), a.mean(), np.percentile(a, 90), np.percentile(a, 99), a.max()))
for x in sorted(times, reverse=True)[:8]:
    print("  %.3f ms  tile (%d, %d)" % x)
