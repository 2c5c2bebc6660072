"""Generates tests/golden/oracle_vectors.npz: outputs of this repository's own CPU
restatement (oracle/, f64 mode = the reference's arithmetic) on fixed inputs, kept
so that a later change to the restatement or to the scene flattener cannot go
unnoticed.  The reference itself has no tests or vectors and cannot be built here,
so these pin the restatement against itself, not against the Rust code.

  the Cornell box's flattened node and primitive arrays (seed 0x5EED0001);
  2048 seeded rays through it: nearest primitive, t, normal, hit point;
  its 24x24 image at 32 samples per pixel, Philox seed 5: per-pixel sums and
  second moments;
  the same for cornell_smoke_scene (glass, mirror, medium) at 16x16, 32 samples.

Run from the repository root:  python tests/golden/make_oracle_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build():
    from conftest import CornellFixture, cornell_ray_batch
    from oracle import binding as orc
    from raytracing_amd import Scene, make_params, scenes

    c = CornellFixture()
    flat = c.flat
    nodes = np.array([[n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2], n.left, n.right]
                      for n in (flat.nodes[i] for i in range(flat.n_nodes))], dtype=np.float64)
    prims = np.array([[q.type, q.material] + [q.f[k] for k in range(6)] for q in (flat.prims[i] for i in range(flat.n_prims))],
                     dtype=np.float64)
    o, d = cornell_ray_batch(c, 2048, seed=123)
    prim, t, normal, p = orc.trace_closest(flat, o, d, mode=orc.ORC_F64, want_normal=True, want_p=True)
    s, s2 = orc.render(flat, c.camera.c, make_params(24, 24, 32, seed=5), mode=orc.ORC_F64, n_threads=1)
    with Scene(0x5EED0001) as sc:
        cam, world = scenes.cornell_smoke_scene(1.0)
        sc.light_xz_rect(*scenes.SMOKE_LIGHT_SHAPE)
        sflat = sc.flatten(world)
        ss, ss2 = orc.render(sflat, cam.c, make_params(16, 16, 32, seed=6), mode=orc.ORC_F64, n_threads=1)
    return dict(nodes=nodes, prims=prims, ray_o=o, ray_d=d, ray_prim=prim, ray_t=t, ray_normal=normal, ray_p=p,
                cornell_sum=np.asarray(s, dtype=np.float64), cornell_sumsq=np.asarray(s2, dtype=np.float64),
                smoke_sum=np.asarray(ss, dtype=np.float64), smoke_sumsq=np.asarray(ss2, dtype=np.float64))


if __name__ == "__main__":
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    v = build()
    np.savez_compressed(dst, **v)
    print("wrote", dst, {k: a.shape for k, a in v.items()}, os.path.getsize(dst), "bytes")
