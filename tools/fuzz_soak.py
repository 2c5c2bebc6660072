#!/usr/bin/env python3
"""A longer run of the randomised nearest-hit parity of tests/test_gpu_fuzz.py: N random scenes (1 to 600 items of every
hitable and nesting the flattener takes), 20,000 rays each, device against the fp32 oracle bit for bit (primitive, t,
normal, point), through the device's own tree, through the given one kept (RT_OPT_KEEP_TREE) and through the device-built one,
and the f64 parity mode against the f64 oracle; every second scene nests its items (Translate / RotateY / HitableList /
BvhNode around each other, up to four levels).  tools/fuzz_soak.py [N [first_seed]]
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
from test_gpu_nesting import _random_nested  # noqa: E402
from raytracing_amd._lib import RT_OPT_KEEP_TREE  # noqa: E402

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
        depth = int(rng.randint(1, 5)) if k % 2 else 0
        if depth:
            n_items = min(n_items, 120)
        items = [_random_nested(rng, mats, depth) if depth else _random_hitable(rng, mats) for _ in range(n_items)]
        flat = sc.flatten(S.HitableList.new(items))
        m = 20000
        o = rng.uniform(-14, 14, size=(m, 3)).astype(np.float32)
        d = (rng.uniform(-10, 10, size=(m, 3)).astype(np.float32) - o).astype(np.float32)
        d[: m // 10, rng.randint(3)] = 0.0
        for name, scene in (("given", flat), ("kept", flat), ("lbvh", None)):
            if scene is None:
                scene = gpu.build_lbvh(flat)
            gpu.set_option(RT_OPT_KEEP_TREE, 1 if name == "kept" else 0)
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
                # A ray that lies exactly in a rect's plane (direction component 0 and origin component equal to the
                # rect's k: t = 0/0) is the one case that is not reproduced: the reference's own answer there depends on
                # how each later comparison treats NaN.  Random floats do coincide like that once in a few billion pairs.
                diff = np.nonzero((prim != oprim) | (t != ot.astype(np.float32)))[0]
                axis_of = {1: 2, 2: 1, 3: 0}  # RT_PRIM_XY_RECT, XZ, YZ -> the axis of k
                planes = {}
                for q in range(scene.n_prims):
                    ty = scene.prims[q].type & 7
                    if ty in axis_of:
                        planes.setdefault(axis_of[ty], set()).add(np.float32(scene.prims[q].f[4]))
                in_plane = [i for i in diff if any(d[i, a] == 0.0 and o[i, a] in planes.get(a, ()) for a in range(3))]
                if len(in_plane) == len(diff):
                    print("seed %d tree %s: %d ray(s) lying exactly in a rect's plane differ (not reproduced by design)" % (seed, name, len(diff)), flush=True)
                    continue
                bad_scenes += 1
                print("DIFFERS seed %d items %d depth %d tree %s: %d primitive ids, %d t values" % (
                    seed, n_items, depth, name, int((prim != oprim).sum()), int((t != ot.astype(np.float32)).sum())), flush=True)
    if (k + 1) % 20 == 0:
        print("... %d scenes, %.0f s" % (k + 1, time.time() - t0), flush=True)
print("%d scenes, %d rays, %d scene/tree pairs differ" % (n_scenes, rays, bad_scenes))
gpu.close()
sys.exit(1 if bad_scenes else 0)
