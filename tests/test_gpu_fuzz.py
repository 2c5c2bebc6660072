"""Randomised nearest-hit parity: scenes assembled at random from every kind of
hitable the flattener takes (spheres, the three rects with and without
FlipNormals, cuboids, Translate / RotateY wrappers in every supported
combination, nested BvhNodes, lists), sized so that all three traversal layouts
come up, against the CPU oracle bit for bit."""
import numpy as np
import pytest

from raytracing_amd import Scene
from raytracing_amd import scene as S

pytestmark = pytest.mark.gpu


def _random_hitable(rng, mats, allow_nested=True):
    m = mats[rng.randint(len(mats))]
    c = rng.uniform(-10, 10, 3)
    kind = rng.randint(8 if allow_nested else 6)
    if kind == 0:
        # one sphere in five has a negative radius: the hollow-glass trick, normals point inwards (hitable.rs:63)
        return S.Sphere.new(S.vec3(*c), float(rng.uniform(0.2, 2.0)) * (-1.0 if rng.rand() < 0.2 else 1.0), m)
    if kind in (1, 2, 3):
        a0, b0 = rng.uniform(-10, 8, 2)
        a1, b1 = a0 + rng.uniform(0.5, 6), b0 + rng.uniform(0.5, 6)
        k = float(rng.uniform(-10, 10))
        make = (S.XYRect.new, S.XZRect.new, S.YZRect.new)[kind - 1]
        flip = (S.FlipNormals.new_xy, S.FlipNormals.new_xz, S.FlipNormals.new_yz)[kind - 1]
        r = make(float(a0), float(a1), float(b0), float(b1), k, m)
        return flip(r) if rng.rand() < 0.5 else r
    if kind in (4, 5):
        inner = S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(*rng.uniform(0.5, 4, 3)), m)
        if not allow_nested:
            return inner  # inside a BvhNode only unwrapped hitables are supported
    elif kind == 6:
        inner = S.BvhNode.new([_random_hitable(rng, mats, False) for _ in range(rng.randint(1, 12))])
    else:
        inner = S.HitableList.new([S.Sphere.new(S.vec3(*rng.uniform(0, 3, 3)), float(rng.uniform(0.2, 0.8)), m) for _ in range(rng.randint(1, 5))])
    wrap = rng.randint(4)
    if kind == 4 and wrap == 0:
        return inner  # a bare cuboid: a six-rect list leaf
    if isinstance(inner, S.BvhNode) and wrap == 0:
        wrap = 3
    if wrap in (1, 3):
        inner = S.RotateY.new(inner, float(rng.uniform(-80, 80)))
    if wrap in (2, 3) or wrap == 0:
        inner = S.Translate.new(inner, S.vec3(*c))
    return inner


@pytest.mark.parametrize("seed,n_items", [(1, 3), (2, 9), (3, 15), (4, 17), (5, 40), (6, 120), (7, 400), (8, 1), (9, 70), (10, 30)])
def test_random_scenes_trace_bit_exact(gpu, orc, seed, n_items):
    rng = np.random.RandomState(1000 + seed)
    with Scene(seed) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        items = [_random_hitable(rng, mats) for _ in range(n_items)]
        flat = sc.flatten(S.HitableList.new(items))
        gpu.upload_scene(flat)
        m = 20000
        o = rng.uniform(-14, 14, size=(m, 3)).astype(np.float32)
        tgt = rng.uniform(-10, 10, size=(m, 3)).astype(np.float32)
        d = (tgt - o).astype(np.float32)
        d[: m // 10, rng.randint(3)] = 0.0  # some rays parallel to an axis plane
        prim, t, nrm, p = gpu.trace_closest(o, d, want_normal=True, want_p=True)
        oprim, ot, onrm, op = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED, want_normal=True, want_p=True)
        bad = np.nonzero(prim != oprim)[0]
        assert bad.size == 0, (seed, bad[:5], prim[bad[:5]], oprim[bad[:5]])
        assert np.array_equal(t, ot.astype(np.float32))
        assert np.array_equal(nrm, onrm.astype(np.float32))
        assert np.array_equal(p, op.astype(np.float32))
        # the same through the device-built tree
        lb = gpu.build_lbvh(flat)
        gpu.upload_scene(lb)
        prim2, t2 = gpu.trace_closest(o, d)
        oprim2, ot2 = orc.trace_closest(lb, o, d, mode=orc.ORC_F32_FUSED)
        assert np.array_equal(prim2, oprim2) and np.array_equal(t2, ot2.astype(np.float32))


@pytest.mark.parametrize("seed,n_items,media", [(21, 6, 1), (22, 14, 2), (23, 40, 1), (24, 150, 0), (25, 10, 0)])
def test_random_scenes_render_tracks_oracle(gpu, orc, seed, n_items, media):
    """Random scenes with every material, constant media in random boxes, a lamp
    and the gradient sky, through the flat and the stack layouts: same random
    stream against the fp32 oracle, independent streams against the f64 oracle."""
    from raytracing_amd import _lib
    from test_gpu_materials import _compare
    rng = np.random.RandomState(2000 + seed)
    with Scene(seed) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(*(0.2 + 0.6 * rng.rand(3)))), S.Metal.new(S.vec3(0.8, 0.7, 0.6), 0.2),
                S.Dielectric.new(1.5), S.Lambertian.new_from_color(S.vec3(0.7, 0.7, 0.7))]
        items = [_random_hitable(rng, mats) for _ in range(n_items)]
        items.append(S.Sphere.new(S.vec3(0.0, -1010.0, 0.0), 1000.0, mats[3]))
        items.append(S.FlipNormals.new_xz(S.XZRect.new(-4.0, 4.0, -4.0, 4.0, 14.0, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(8.0))))))
        for _ in range(media):
            box = S.Translate.new(S.RotateY.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(*rng.uniform(3, 8, 3)), mats[0]), float(rng.uniform(-40, 40))),
                                  S.vec3(*rng.uniform(-8, 2, 3)))
            items.append(S.ConstantMedium.new(box, float(rng.uniform(0.05, 0.3)), S.Texture.new_constant(S.vec3(*rng.rand(3)))))
        sc.light_xz_rect(-4.0, 4.0, -4.0, 4.0, 14.0)
        cam = S.Camera.new(S.vec3(0.0, 6.0, -34.0), S.vec3(0.0, 0.0, 0.0), S.vec3(0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0)
        flat = sc.flatten(S.HitableList.new(items))
        _compare(gpu, orc, flat, cam, 48, 32, 48, sky=_lib.RT_SKY_GRADIENT, same_frac=0.985, tight_frac=0.95, energy_tol=0.005)
