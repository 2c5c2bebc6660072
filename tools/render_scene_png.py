#!/usr/bin/env python3
"""Renders one named scene to a PNG (on a GPU box): tools/render_scene_png.py <scene> <nx> <ny> <spp> <out.png>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (brings up the HIP runtime the library shares)

torch.cuda.set_device(0)
from raytracing_amd import Scene, render_image, scenes  # noqa: E402

name, nx, ny, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
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
    elif name == "stress":
        cam, world = scenes.sphere_stress_scene(ratio, 1 << 16)
        sc.light_xz_rect(*scenes.STRESS_LIGHT_SHAPE)
    else:
        cam, world = scenes.cornell_box_scene(ratio)
        sc.light_from_world(world)
    render_image(sc, cam, world, nx, ny, spp, png_path=out)
print("wrote", out)
