#!/usr/bin/env python3
"""What the final scene's media and its noise texture cost a launch: the same frame with and without each.
tools/final_parts.py [spp]  (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.set_device(0)
from raytracing_amd import Renderer, Scene, make_params, scenes  # noqa: E402

nx, ny = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
buf = torch.empty(ny * nx * 3, dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream(dev)
for marble, media in ((True, True), (True, False), (False, True), (False, False)):
    with Scene(0x5EED0000) as sc:
        cam, world = scenes.final_scene(float(nx) / ny, marble=marble, media=media)
        sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        flat = sc.flatten(world)
    r = Renderer(0)
    r.upload_scene(flat)
    best = 1e30
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        r.render_device(cam, make_params(nx, ny, spp, seed=2), buf.data_ptr(), None, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1))
    _, st = r.render(cam, make_params(nx // 4, ny // 4, 16, seed=2), want_stats=True)
    print("marble %-5s media %-5s  %.2f ms   rays/path %.3f" % (marble, media, best, st.rays / float(st.paths)), flush=True)
    r.close()
