"""Every nesting the constructors allow (hitable.rs:433,465 take any Box<dyn Hitable>):
Translate and RotateY in any order and number, around primitives, lists, BvhNodes
and each other; BvhNodes whose leaves are such things; a ConstantMedium bounded by a
BvhNode.  The flattener writes them as objects inside objects (include/rt_abi.h),
the stack traversal walks them with one frame marker per level, and the hit is
carried back to the world through the chain of objects: nearest hit, point and
normal bit for bit against the fp32 oracle, in the reference's own arithmetic
against the f64 oracle, and renders within the statistical bars."""
import numpy as np
import pytest

from raytracing_amd import Scene, _lib, make_params
from raytracing_amd import scene as S
from test_gpu_fuzz import _random_hitable

pytestmark = pytest.mark.gpu


def _rays(rng, m, lo=-16.0, hi=16.0, aim=8.0):
    o = rng.uniform(lo, hi, size=(m, 3)).astype(np.float32)
    tgt = rng.uniform(-aim, aim, size=(m, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    d[: m // 10, rng.randint(3)] = 0.0
    return o, d


def _check_trace(gpu, orc, flat, rng, m=20000):
    gpu.upload_scene(flat)
    o, d = _rays(rng, m)
    prim, t, nrm, p = gpu.trace_closest(o, d, want_normal=True, want_p=True)
    oprim, ot, onrm, op = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED, want_normal=True, want_p=True)
    assert (prim >= 0).mean() > 0.05
    assert np.array_equal(prim, oprim)
    assert np.array_equal(t, ot.astype(np.float32))
    assert np.array_equal(nrm, onrm.astype(np.float32))
    assert np.array_equal(p, op.astype(np.float32))
    # the reference's own arithmetic on the device against the f64 restatement
    o64, d64 = o[:4000].astype(np.float64), d[:4000].astype(np.float64)
    prim64, t64, n64, p64 = gpu.trace_closest_f64(o64, d64, want_normal=True, want_p=True)
    qprim, qt, qn, qp = orc.trace_closest(flat, o64, d64, mode=orc.ORC_F64, want_normal=True, want_p=True)
    assert np.array_equal(prim64, qprim) and np.array_equal(t64, qt) and np.array_equal(n64, qn) and np.array_equal(p64, qp)
    return prim


def test_rotate_of_translate_and_deeper_chains(gpu, orc):
    """RotateY(Translate(Cuboid)), Translate(RotateY(Translate(Cuboid))), RotateY(RotateY(sphere list)),
    Translate(Translate(rect)): one object per Translate/RotateY pair, each in a sub-tree of one leaf of the one around it."""
    rng = np.random.RandomState(5)
    with Scene(41) as sc:
        m = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        box = lambda k: S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(2.0 + k, 3.0, 1.5 + k), m[k % 3])  # noqa: E731
        items = [
            S.RotateY.new(S.Translate.new(box(0), S.vec3(3.0, -1.0, 2.0)), 25.0),
            S.Translate.new(S.RotateY.new(S.Translate.new(box(1), S.vec3(-2.0, 0.5, 1.0)), -40.0), S.vec3(-5.0, 0.0, 3.0)),
            S.RotateY.new(S.RotateY.new(S.HitableList.new([S.Sphere.new(S.vec3(4.0, 2.0, 0.0), 1.0, m[0]), S.Sphere.new(S.vec3(5.5, 2.0, 0.5), 0.7, m[1])]), 30.0), 15.0),
            S.Translate.new(S.Translate.new(S.XZRect.new(-2.0, 2.0, -2.0, 2.0, 0.0, m[2]), S.vec3(0.0, -4.0, 0.0)), S.vec3(1.0, 0.0, 1.0)),
            S.Sphere.new(S.vec3(0.0, -1010.0, 0.0), 1000.0, m[2]),
        ]
        flat = sc.flatten(S.HitableList.new(items))
        kinds = [flat.objects[i].kind for i in range(flat.n_objects)]
        assert flat.n_objects == 8 and kinds.count(_lib.RT_OBJ_BVH) == 4 and kinds.count(_lib.RT_OBJ_LIST) == 4
        prim = _check_trace(gpu, orc, flat, rng)
        assert len(np.unique(prim)) > 12  # every item is hit somewhere


def test_instanced_subtree_that_holds_instances(gpu, orc):
    """Translate(RotateY(BvhNode([RotateY(Cuboid), Translate(sphere list), BvhNode([...]) ...]))): frames two and three deep
    inside a tree, plain and instanced leaves side by side."""
    rng = np.random.RandomState(6)
    with Scene(42) as sc:
        m = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        inner = []
        for k in range(14):
            c = rng.uniform(-6, 6, 3)
            if k % 3 == 0:
                inner.append(S.RotateY.new(S.Cuboid.new(S.vec3(*c), S.vec3(*(c + rng.uniform(0.5, 2.5, 3))), m[k % 3]), float(rng.uniform(-60, 60))))
            elif k % 3 == 1:
                inner.append(S.Translate.new(S.HitableList.new([S.Sphere.new(S.vec3(0, 0, 0), 0.8, m[0]), S.Sphere.new(S.vec3(1.0, 0.3, 0.0), 0.5, m[1])]), S.vec3(*c)))
            else:
                inner.append(S.Sphere.new(S.vec3(*c), float(rng.uniform(0.3, 1.2)), m[k % 3]))
        inner.append(S.Translate.new(S.BvhNode.new([S.RotateY.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(1, 2, 1), m[1]), 33.0),
                                                    S.Sphere.new(S.vec3(2.0, 1.0, 0.0), 0.6, m[2])]), S.vec3(-3.0, 4.0, -2.0)))
        items = [S.Translate.new(S.RotateY.new(S.BvhNode.new(inner), 20.0), S.vec3(1.0, 0.0, -1.0)),
                 S.BvhNode.new([S.Translate.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(2, 1, 2), m[0]), S.vec3(9.0, -3.0, 0.0)),
                                S.Sphere.new(S.vec3(-9.0, 3.0, 0.0), 1.5, m[1])]),
                 S.Sphere.new(S.vec3(0.0, -1012.0, 0.0), 1000.0, m[2])]
        flat = sc.flatten(S.HitableList.new(items))
        _check_trace(gpu, orc, flat, rng)


def _random_nested(rng, mats, depth):
    """Random hitables with wrappers in any order: up to `depth` levels of Translate / RotateY / HitableList / BvhNode around
    the fuzzer's plain hitables."""
    if depth == 0 or rng.rand() < 0.25:
        return _random_hitable(rng, mats, allow_nested=rng.rand() < 0.5)
    kind = rng.randint(4)
    if kind == 0:
        return S.Translate.new(_random_nested(rng, mats, depth - 1), S.vec3(*rng.uniform(-4, 4, 3)))
    if kind == 1:
        return S.RotateY.new(_random_nested(rng, mats, depth - 1), float(rng.uniform(-90, 90)))
    if kind == 2:
        return S.HitableList.new([_random_nested(rng, mats, depth - 1) for _ in range(rng.randint(1, 4))])
    return S.BvhNode.new([_random_nested(rng, mats, depth - 1) for _ in range(rng.randint(1, 6))])


@pytest.mark.parametrize("seed,n_items,depth", [(1, 4, 2), (2, 10, 3), (3, 25, 3), (4, 2, 4), (5, 60, 2), (6, 8, 4)])
def test_random_nestings_trace_bit_exact(gpu, orc, seed, n_items, depth):
    rng = np.random.RandomState(3000 + seed)
    with Scene(seed) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(*rng.rand(3))) for _ in range(3)]
        items = [_random_nested(rng, mats, depth) for _ in range(n_items)]
        flat = sc.flatten(S.HitableList.new(items))
        _check_trace(gpu, orc, flat, rng, m=12000)


def test_nested_scene_renders_like_the_oracle(gpu, orc):
    """The chains above in a lit room: same random stream against the fp32 oracle, independent streams against f64."""
    from test_gpu_materials import _compare
    rng = np.random.RandomState(9)
    with Scene(43) as sc:
        grey = S.Lambertian.new_from_color(S.vec3(0.7, 0.7, 0.7))
        red = S.Lambertian.new_from_color(S.vec3(0.7, 0.2, 0.2))
        glass = S.Dielectric.new(1.5)
        items = [
            S.Sphere.new(S.vec3(0.0, -1000.0, 0.0), 1000.0, grey),
            S.RotateY.new(S.Translate.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(2, 3, 2), red), S.vec3(2.0, 0.0, 1.0)), 30.0),
            S.Translate.new(S.RotateY.new(S.Translate.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(1.5, 1.5, 1.5), grey), S.vec3(0.0, 0.0, 1.0)), -25.0), S.vec3(-4.0, 0.0, 0.0)),
            S.Translate.new(S.RotateY.new(S.BvhNode.new([S.RotateY.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(1, 1, 1), red), 45.0),
                                                         S.Translate.new(S.Sphere.new(S.vec3(0, 0, 0), 0.7, glass), S.vec3(2.0, 0.7, 0.0)),
                                                         S.Sphere.new(S.vec3(-1.5, 0.5, 0.5), 0.5, grey)]), 10.0), S.vec3(0.0, 0.0, -3.0)),
            S.FlipNormals.new_xz(S.XZRect.new(-3.0, 3.0, -3.0, 3.0, 8.0, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(7.0))))),
        ]
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 8.0)
        cam = S.Camera.new(S.vec3(0.0, 4.0, -14.0), S.vec3(0.0, 1.0, 0.0), S.vec3(0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0)
        flat = sc.flatten(S.HitableList.new(items))
        _compare(gpu, orc, flat, cam, 60, 40, 64, sky=_lib.RT_SKY_GRADIENT)


def test_medium_bounded_by_a_bvh_node(gpu, orc):
    """ConstantMedium::new takes any boundary (hitable.rs:551): here a BvhNode of spheres and a rotated cuboid under
    Translate, i.e. two boundary searches through a sub-tree with frames of its own."""
    from test_gpu_materials import _compare
    with Scene(44) as sc:
        grey = S.Lambertian.new_from_color(S.vec3(0.7, 0.7, 0.7))
        cloud = S.BvhNode.new([S.Sphere.new(S.vec3(0.0, 2.0, 0.0), 2.0, grey), S.Sphere.new(S.vec3(2.5, 2.0, 0.5), 1.5, grey),
                               S.RotateY.new(S.Cuboid.new(S.vec3(-4.0, 0.5, -1.0), S.vec3(-2.0, 3.0, 1.0), grey), 20.0)])
        items = [
            S.Sphere.new(S.vec3(0.0, -1000.0, 0.0), 1000.0, grey),
            S.ConstantMedium.new(S.Translate.new(cloud, S.vec3(0.0, 0.0, 1.0)), 0.4, S.Texture.new_constant(S.vec3(0.9, 0.9, 0.95))),
            S.ConstantMedium.new(S.BvhNode.new([S.Sphere.new(S.vec3(5.0, 1.0, -2.0), 1.0, grey)]), 0.8, S.Texture.new_constant(S.vec3(0.2, 0.4, 0.9))),
            S.Sphere.new(S.vec3(-5.0, 1.0, -2.0), 1.0, S.Lambertian.new_from_color(S.vec3(0.8, 0.3, 0.2))),
            S.FlipNormals.new_xz(S.XZRect.new(-3.0, 3.0, -3.0, 3.0, 9.0, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(7.0))))),
        ]
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 9.0)
        cam = S.Camera.new(S.vec3(0.0, 4.0, -16.0), S.vec3(0.0, 1.5, 0.0), S.vec3(0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0)
        flat = sc.flatten(S.HitableList.new(items))
        kinds = [flat.objects[i].kind for i in range(flat.n_objects)]
        assert kinds.count(_lib.RT_OBJ_MEDIUM_BVH) == 2
        _compare(gpu, orc, flat, cam, 60, 40, 64, sky=_lib.RT_SKY_GRADIENT)
        # the media are invisible to a deterministic query
        rng = np.random.RandomState(3)
        _check_trace(gpu, orc, flat, rng, m=6000)


def test_a_medium_inside_another_hitable_is_refused():
    from raytracing_amd import RtError
    with Scene(45) as sc:
        grey = S.Lambertian.new_from_color(S.vec3(0.7, 0.7, 0.7))
        fog = S.ConstantMedium.new(S.Sphere.new(S.vec3(0, 0, 0), 1.0, grey), 0.5, S.Texture.new_constant(S.vec3(1, 1, 1)))
        with pytest.raises(RtError) as e:
            sc.flatten(S.HitableList.new([S.Translate.new(fog, S.vec3(1.0, 0.0, 0.0)), S.Sphere.new(S.vec3(3, 0, 0), 1.0, grey)]))
        assert e.value.code == _lib.RT_ERR_UNSUPPORTED


def test_keep_tree_option_walks_the_given_interior(gpu, orc):
    """RT_OPT_KEEP_TREE: the device collapses the interior nodes it is given instead of building its own over the leaves.
    Any tree over the same leaves gives the same nearest hits, so the traces stay bit-exact against the oracle (which walks
    the given tree) for the flattener's tree, for nested objects, and for the tree rt_build_lbvh makes on the device; the
    number of box tests shows that another tree is being walked."""
    from raytracing_amd._lib import RT_OPT_KEEP_TREE
    rng = np.random.RandomState(77)
    with Scene(5) as sc:
        mats = [S.Lambertian.new_from_color(S.vec3(0.6, 0.5, 0.4))]
        world = S.HitableList.new([_random_nested(rng, mats, 2) for _ in range(150)])
        flat = sc.flatten(world)
    p = make_params(64, 48, 4, seed=3)
    cam = S.Camera.new(S.vec3(0.0, 4.0, 30.0), S.vec3(0.0, 0.0, 0.0), S.vec3(0.0, 1.0, 0.0), 40.0, 64.0 / 48.0, 0.0, 10.0)
    gpu.set_option(RT_OPT_KEEP_TREE, 0)  # (whatever the environment preset)
    gpu.upload_scene(flat)
    _, st_own = gpu.render(cam, p, want_stats=True)
    gpu.set_option(RT_OPT_KEEP_TREE, 1)
    try:
        prim = _check_trace(gpu, orc, flat, rng, m=12000)
        assert (prim >= 0).sum() > 500
        _, st_kept = gpu.render(cam, p, want_stats=True)
        assert (st_kept.rays, st_kept.hits) == (st_own.rays, st_own.hits)
        assert st_kept.node_tests != st_own.node_tests
        # a world of plain spheres through the device-built linear BVH, kept
        from test_gpu_trace import _lbvh_scene
        sc2, flat2 = _lbvh_scene(3000, 11)
        lb = gpu.build_lbvh(flat2)
        gpu.upload_scene(lb)
        o, d = _rays(rng, 12000, lo=-30.0, hi=30.0, aim=10.0)
        prim_l, t_l = gpu.trace_closest(o, d)
        oprim, ot = orc.trace_closest(lb, o, d, mode=orc.ORC_F32_FUSED)  # the oracle walks the same device-built tree
        assert (prim_l >= 0).mean() > 0.02
        assert np.array_equal(prim_l, oprim) and np.array_equal(t_l, ot.astype(np.float32))
    finally:
        gpu.set_option(RT_OPT_KEEP_TREE, 0)


def test_sphere_of_negative_radius_in_a_wrapped_list(gpu, orc):
    """A sphere of negative radius (the hollow-sphere trick) has an inverted box of its own (hitable.rs:86-91), so the boxes
    above a list that holds one do not contain it and a ray can hit it before it enters them.  The reference walks every
    box the ray passes whatever has been found already, so it finds that hit; the device must not skip such a box for
    lying beyond its best hit so far."""
    rng = np.random.RandomState(5)
    with Scene(9) as sc:
        m = S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5))
        items = [S.Sphere.new(S.vec3(*rng.uniform(-14, 14, 3)), float(rng.uniform(0.3, 1.5)), m) for _ in range(120)]
        for k in range(12):
            c = rng.uniform(-8, 8, 3)
            inner = S.HitableList.new([S.Sphere.new(S.vec3(*c), -float(rng.uniform(1.0, 2.5)), m),
                                       S.Sphere.new(S.vec3(*(c + rng.uniform(-3, 3, 3))), 0.4, m)])
            items.append(S.Translate.new(S.RotateY.new(inner, float(rng.uniform(-60, 60))), S.vec3(*rng.uniform(-1, 1, 3))))
        # the same inside a list inside a turned list: the inner list's box (too small) is the only box between a ray
        # and the hollow sphere, and it must stay one (an untransformed nested list is not merged into its parent's tree)
        for k in range(8):
            c = rng.uniform(-8, 8, 3)
            inner = S.HitableList.new([S.Sphere.new(S.vec3(*c), -float(rng.uniform(1.0, 2.5)), m),
                                       S.Sphere.new(S.vec3(*(c + rng.uniform(-3, 3, 3))), 0.4, m)])
            items.append(S.RotateY.new(S.HitableList.new([inner, S.Sphere.new(S.vec3(*rng.uniform(-8, 8, 3)), 0.5, m)]), float(rng.uniform(-60, 60))))
        # and a BvhNode of one hollow sphere inside a turned list: its box is the sphere's inverted one, which no ray passes
        # (the tree under the list then is a single leaf, whose own box is the one to test on the way in)
        never = []
        for k in range(8):
            c = rng.uniform(-8, 8, 3)
            items.append(S.RotateY.new(S.HitableList.new([S.BvhNode.new([S.Sphere.new(S.vec3(*c), -2.0, m)])]), float(rng.uniform(-60, 60))))
            never.append(c)
        flat = sc.flatten(S.HitableList.new(items))
    prim = _check_trace(gpu, orc, flat, rng, m=40000)
    hollow = [k for k in range(flat.n_prims) if flat.prims[k].f[3] < 0.0]
    assert len(hollow) == 28 and np.isin(prim, hollow).sum() > 100
    unreachable = [k for k in hollow if flat.prims[k].f[3] == -2.0]
    assert len(unreachable) == 8 and not np.isin(prim, unreachable).any()
