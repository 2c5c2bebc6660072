#!/usr/bin/env python3
"""A longer run of the randomised nearest-hit parity of tests/test_gpu_fuzz.py: N random scenes (1 to 600 items of every
hitable and nesting the flattener takes), 20,000 rays each, device against the fp32 oracle bit for bit (primitive, t,
normal, point), through the tree as given and through the device-built one, and the f64 parity mode against the f64 oracle.  tools/fuzz_soak.py [N [first_seed]]
(on the GPU box; prints one line per scene that differs and a summary)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from raytracing_amd import Renderer, Scene  # noqa: E402
from raytracing_amd import scene as S  # noqa: E402
from oracle import binding as orc  # noqa: E402
from test_gpu_fuzz import _random_hitable  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
gpu = Renderer(0)
bad_scenes = 0
rays = 0
layouts = {}
t0 = time.time()
for k in range(n_scenes):
    seed = first + k
    rng = np.random.RandomState(seed)
    n_items = int(rng.choice([1, 2, 5, 12, 17, 30, 60, 120, 300, 600]))
    with Scene(seed) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        items = [_random_hitable(rng, mats) for _ in range(n_items)]
        flat = sc.flatten(S.HitableList.new(items))
        m = 20000
        o = rng.uniform(-14, 14, size=(m, 3)).astype(np.float32)
        d = (rng.uniform(-10, 10, size=(m, 3)).astype(np.float32) - o).astype(np.float32)
        d[: m // 10, rng.randint(3)] = 0.0
        for name, scene in (("given", flat), ("lbvh", None)):
            if scene is None:
                scene = gpu.build_lbvh(flat)
            gpu.upload_scene(scene)
            prim, t, nrm, p = gpu.trace_closest(o, d, want_normal=True, want_p=True)
            oprim, ot, onrm, op = orc.trace_closest(scene, o, d, mode=orc.ORC_F32_FUSED, want_normal=True, want_p=True)
            ok = (np.array_equal(prim, oprim) and np.array_equal(t, ot.astype(np.float32)) and
                  np.array_equal(nrm, onrm.astype(np.float32)) and np.array_equal(p, op.astype(np.float32)))
            if name == "given":
                # the parity mode: the device in the reference's own arithmetic against the f64 oracle
                p64, t64, n64, x64 = gpu.trace_closest_f64(o.astype(np.float64), d.astype(np.float64), want_normal=True, want_p=True)
                q64, u64, m64, y64 = orc.trace_closest(scene, o, d, mode=orc.ORC_F64, want_normal=True, want_p=True)
                if not (np.array_equal(p64, q64) and np.array_equal(t64, u64) and np.array_equal(n64, m64) and np.array_equal(x64, y64)):
                    ok = False
                    print("  (f64 parity mode differs: %d primitive ids, %d t values)" % (int((p64 != q64).sum()), int((t64 != u64).sum())), flush=True)
                rays += m
            rays += m
            if not ok:
                bad_scenes += 1
                print("DIFFERS seed %d items %d tree %s: %d primitive ids, %d t values" % (
                    seed, n_items, name, int((prim != oprim).sum()), int((t != ot.astype(np.float32)).sum())), flush=True)
    if (k + 1) % 20 == 0:
        print("... %d scenes, %.0f s" % (k + 1, time.time() - t0), flush=True)
print("%d scenes, %d rays, %d scene/tree pairs differ" % (n_scenes, rays, bad_scenes))
gpu.close()
sys.exit(1 if bad_scenes else 0)
