#!/usr/bin/env python3
"""What one rank's share of the headline image costs: the Cornell box at 1024x1024 with 125, 250 and 500 of the 1000 samples
(a rank's launch at N = 8, 4, 2) against the whole image, for several floors of the sample-chunk size and item counts
(RT_MIN_CHUNK_SPP, RT_WANT_ROUNDS).  tools/cornell_shard_cost.py  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
from raytracing_amd import Renderer, Scene, make_params, scenes
with Scene(0x5EED0000) as sc:
    cam, world = scenes.cornell_box_scene(1.0)
    sc.light_xz_rect(*scenes.MAIN_RS_LIGHT_SHAPE)
    flat = sc.flatten(world)
r = Renderer(0); r.upload_scene(flat)
dev = torch.device("cuda", 0)
buf = torch.empty(1024 * 1024 * 3, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
def t(p):
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); r.render_device(cam, p, buf.data_ptr(), None, stream.cuda_stream); e1.record(stream)
        torch.cuda.synchronize(dev); best = min(best, e0.elapsed_time(e1))
    return best
whole = t(make_params(1024, 1024, 1000, seed=1))
for m in ("32", "16", "8"):
    for rounds in ("30", "60"):
        os.environ["RT_MIN_CHUNK_SPP"] = m; os.environ["RT_WANT_ROUNDS"] = rounds
        a = t(make_params(1024, 1024, 125, seed=1)); b = t(make_params(1024, 1024, 250, seed=1)); c = t(make_params(1024, 1024, 500, seed=1))
        print("min %s rounds %s: 125 spp %.3f ms (x8 = %.2f, eff %.3f)  250 spp %.3f (eff %.3f)  500 spp %.3f (eff %.3f)" % (m, rounds, a, 8 * a, whole / (8 * a), b, whole / (4 * b), c, whole / (2 * c)), flush=True)
print("whole 1000 spp %.3f" % whole)
