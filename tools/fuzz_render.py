#!/usr/bin/env python3
"""Randomised render parity: N random scenes (plain and nested items, every material, checker and noise textures, an emitter
or two, sometimes a ConstantMedium, gradient sky), rendered small on the device and by the fp32 oracle with the same random
stream; prints the share of pixel sums that agree and flags scenes below the bar.  tools/fuzz_render.py [N [first_seed]]
(on the GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from raytracing_amd import Renderer, Scene, _lib, make_params  # noqa: E402
from raytracing_amd import scene as S  # noqa: E402
from oracle import binding as orc  # noqa: E402
from test_gpu_fuzz import _random_hitable  # noqa: E402
from test_gpu_nesting import _random_nested  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
gpu = Renderer(0)
bad = 0
worst = 1.0
t0 = time.time()
nx, ny, spp = 48, 32, 8
for k in range(n_scenes):
    seed = first + k
    rng = np.random.RandomState(seed)
    n_items = int(rng.choice([3, 8, 20, 40, 90]))
    with Scene(seed) as sc:
        tex_noise = S.Texture.new_noise(float(rng.uniform(0.5, 4.0)))
        tex_check = S.Texture.new_checker(S.Texture.new_constant(S.vec3(0.2, 0.3, 0.1)), S.Texture.new_constant(S.vec3(0.9, 0.9, 0.9)))
        mats = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))), S.Metal.new(S.vec3(*(0.5 + 0.5 * rng.rand(3))), float(rng.rand() * 0.5)),
                S.Dielectric.new(1.5), S.Lambertian.new(tex_noise), S.Lambertian.new(tex_check),
                S.DiffuseLight.new(S.Texture.new_constant(S.scalar(4.0)))]
        depth = int(rng.randint(1, 4)) if k % 2 else 0
        items = []
        for _ in range(n_items):
            pick = [mats[int(rng.randint(len(mats)))] for _ in range(3)]
            items.append(_random_nested(rng, pick, depth) if depth else _random_hitable(rng, pick))
        if rng.rand() < 0.4:
            c = rng.uniform(-5, 5, 3)
            items.append(S.ConstantMedium.new(S.Sphere.new(S.vec3(*c), float(rng.uniform(1.5, 4.0)), mats[0]), float(rng.uniform(0.05, 0.5)),
                                              S.Texture.new_constant(S.vec3(*rng.rand(3)))))
        world = S.HitableList.new(items)
        if rng.rand() < 0.5:
            sc.lights_from_world(world)
        flat = sc.flatten(world)
        cam = S.Camera.new(S.vec3(*rng.uniform(-3, 3, 3)) + S.vec3(0.0, 2.0, 26.0), S.vec3(0.0, 0.0, 0.0), S.vec3(0.0, 1.0, 0.0), 45.0, float(nx) / ny,
                           0.0 if rng.rand() < 0.5 else 0.2, 26.0)
        p = make_params(nx, ny, spp, seed=seed, sky=_lib.RT_SKY_GRADIENT, max_depth=int(rng.choice([5, 50])))
        gpu.upload_scene(flat)
        s = gpu.render(cam, p)
        o = orc.render(flat, cam.c, p, mode=orc.ORC_F32_FUSED, want_sumsq=False, n_threads=8)
        close = float(np.isclose(s, o, rtol=5e-3, atol=5e-3 * spp).mean())
        energy = abs(float(s.sum()) - float(o.sum())) / max(float(o.sum()), 1e-9)
        worst = min(worst, close)
        if not np.isfinite(s).all() or close < 0.97 or energy > 0.02:
            bad += 1
            print("BELOW THE BAR seed %d items %d depth %d: %.4f of the sums agree, energy off by %.4f" % (seed, n_items, depth, close, energy), flush=True)
    if (k + 1) % 10 == 0:
        print("... %d scenes, %.0f s, worst share so far %.4f" % (k + 1, time.time() - t0, worst), flush=True)
print("%d scenes, %d below the bar, worst share of agreeing sums %.4f" % (n_scenes, bad, worst))
gpu.close()
sys.exit(1 if bad else 0)
