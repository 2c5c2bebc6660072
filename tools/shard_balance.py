#!/usr/bin/env python3
"""Load balance of the two sharding modes, measured on one GPU: every shard of an N-way split is rendered on its own and timed;
the slowest shard bounds the N-GPU time (exchange not included).  tools/shard_balance.py <scene> <nx> <ny> <spp>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.set_device(0)
from raytracing_amd import Renderer, Scene, make_params, scenes  # noqa: E402
from raytracing_amd.distributed import shard_params  # noqa: E402

name, nx, ny, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
with Scene(0x5EED0000) as sc:
    ratio = float(nx) / ny
    if name == "final":
        cam, world = scenes.final_scene(ratio)
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
dev = torch.device("cuda", 0)
buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)


def timed(p):
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best


whole = timed(make_params(nx, ny, spp))
out = {"scene": name, "nx": nx, "ny": ny, "spp": spp, "one_gpu_ms": round(whole, 3)}
for mode in ("tiles", "samples"):
    for n in (2, 4, 8):
        if mode == "samples" and spp < n:
            continue
        t = [timed(shard_params(make_params(nx, ny, spp), k, n, mode)) for k in range(n)]
        out["%s_%d" % (mode, n)] = {"slowest_ms": round(max(t), 3), "fastest_ms": round(min(t), 3),
                                    "efficiency_bound": round(whole / (n * max(t)), 3)}
print(json.dumps(out))
r.close()
