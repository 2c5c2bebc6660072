"""Host-side logic: scene construction, BVH build, flattening, camera, and the
C ABI surface.  No GPU needed."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from raytracing_amd import RtError, Scene, _lib
from raytracing_amd import scene as S
from raytracing_amd.scenes import MAIN_RS_LIGHT_SHAPE, cornell_box_scene


def test_library_exports_every_declared_symbol():
    """Every function declared in include/*.h is exported by librt_amd.so."""
    lib = C.CDLL(_lib.LIB_PATH)
    names = set()
    for hdr in ("rt_abi.h", "rt_scene_api.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(rtb?_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 30
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_struct_sizes_match_the_header():
    assert C.sizeof(_lib.rt_node) == 32
    assert C.sizeof(_lib.rt_prim) == 32
    assert C.sizeof(_lib.rt_object) == 48
    assert C.sizeof(_lib.rt_material) == 32
    assert C.sizeof(_lib.rt_texture) == 32
    assert C.sizeof(_lib.rt_light) == 32
    assert C.sizeof(_lib.rt_perlin) == 256 * 3 * 4 + 3 * 256 * 4


def test_cornell_flattening(cornell):
    f = cornell.flat
    # 8 world items -> 15 nodes; 6 wall/light rects + 2 cuboids x 6 rects; 2 instanced objects
    assert (f.n_nodes, f.n_prims, f.n_objects, f.n_materials, f.n_lights) == (15, 18, 2, 4, 1)
    leaves = [f.nodes[i] for i in range(f.n_nodes) if f.nodes[i].left < 0]
    assert len(leaves) == 8
    assert sorted(n.right for n in leaves) == [0, 0, 1, 1, 1, 1, 1, 1]
    # breadth-first: children always come later, every node but the root has one parent
    parents = {}
    for i in range(f.n_nodes):
        n = f.nodes[i]
        if n.left >= 0:
            assert n.left > i and n.right > i
            parents.setdefault(n.left, []).append(i)
            parents.setdefault(n.right, []).append(i)
    assert sorted(parents) == list(range(1, 15)) and all(len(v) == 1 for v in parents.values())
    # parent boxes contain child boxes
    for i in range(f.n_nodes):
        n = f.nodes[i]
        if n.left >= 0:
            for c in (n.left, n.right):
                ch = f.nodes[c]
                assert all(n.bmin[k] <= ch.bmin[k] and n.bmax[k] >= ch.bmax[k] for k in range(3))
    prims = [f.prims[i] for i in range(f.n_prims)]
    # the left wall keeps the source's z range of 565 (scenes.rs:285) and is flipped
    left = [p for p in prims if (p.type & 0xFF) == _lib.RT_PRIM_YZ_RECT and p.f[4] == 555.0 and p.f[3] == 565.0]
    assert len(left) == 1 and (left[0].type & _lib.RT_PRIM_FLIP)
    # each cuboid is its 6 rects in source order with the top face lifted by 0.01 (hitable.rs:382-415)
    for oi in range(f.n_objects):
        ob = f.objects[oi]
        assert ob.kind == _lib.RT_OBJ_LIST and ob.count == 6
        assert ob.xform == (_lib.RT_XFORM_ROTATE_Y | _lib.RT_XFORM_TRANSLATE)
        six = prims[ob.first:ob.first + 6]
        assert [p.type & 0xFF for p in six] == [1, 1, 2, 2, 3, 3]
        assert [bool(p.type & _lib.RT_PRIM_FLIP) for p in six] == [False, True, False, True, False, True]
        height = six[0].f[3]  # y1 of the XY rect
        assert six[2].f[4] == pytest.approx(height + 0.01, abs=1e-4) and six[2].f[4] > height
        assert six[3].f[4] == 0.0
    angles = sorted(math.degrees(math.atan2(f.objects[i].sin_theta, f.objects[i].cos_theta)) for i in range(2))
    assert angles == pytest.approx([-18.0, 15.0], abs=1e-4)


def test_rotate_y_bounding_box_matches_hand_computation():
    # hitable.rs:477-497 for Cuboid(0, 165) rotated by -18 degrees then translated
    with Scene(1) as sc:
        white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
        item = S.Translate.new(S.RotateY.new(S.Cuboid.new(S.scalar(0.0), S.scalar(165.0), white), -18.0),
                               S.vec3(130.0, 0.0, 65.0))
        flat = sc.flatten(S.HitableList.new([item]))
        assert flat.n_nodes == 1
        n = flat.nodes[0]
        s, c = math.sin(math.radians(-18.0)), math.cos(math.radians(-18.0))
        xs = [c * x + s * z for x in (0, 165) for z in (0, 165)]
        zs = [-s * x + c * z for x in (0, 165) for z in (0, 165)]
        assert n.bmin[0] == pytest.approx(min(xs) + 130, rel=1e-6) and n.bmax[0] == pytest.approx(max(xs) + 130, rel=1e-6)
        assert n.bmin[2] == pytest.approx(min(zs) + 65, rel=1e-6) and n.bmax[2] == pytest.approx(max(zs) + 65, rel=1e-6)
        assert (n.bmin[1], n.bmax[1]) == (0.0, 165.0)  # the lifted top face is outside its own box (hitable.rs:375,396)


def test_bvh_build_is_seeded_and_median_split():
    def build(seed, n):
        with Scene(seed) as sc:
            m = S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5))
            rng = np.random.RandomState(5)
            items = [S.Sphere.new(S.vec3(*rng.uniform(-10, 10, 3)), 0.5, m) for _ in range(n)]
            f = sc.flatten(S.HitableList.new(items))
            return [(f.nodes[i].left, f.nodes[i].right, tuple(f.nodes[i].bmin)) for i in range(f.n_nodes)], f.n_prims
    a, npr = build(3, 37)
    b, _ = build(3, 37)
    c, _ = build(4, 37)
    assert a == b and a != c
    assert len(a) == 2 * 37 - 1 and npr == 37
    # one hitable per leaf (bvh.rs:117-125)
    assert sum(1 for l, r, _ in a if l < 0) == 37 and all(r == 1 for l, r, _ in a if l < 0)


def test_camera_matches_camera_rs():
    cam = S.Camera.new(S.vec3(278, 278, -800), S.vec3(278, 278, 0), S.vec3(0, 1, 0), 40.0, 1.0, 0.0, 10.0).c
    hh = math.tan(math.radians(40.0) / 2.0)
    assert list(cam.w) == pytest.approx([0, 0, -1])
    assert list(cam.u) == pytest.approx([-1, 0, 0])  # vup x w with the sign convention of vec3.rs:180-186
    assert list(cam.v) == pytest.approx([0, 1, 0])
    assert list(cam.horizontal) == pytest.approx([-2 * hh * 10, 0, 0])
    assert list(cam.vertical) == pytest.approx([0, 2 * hh * 10, 0])
    assert list(cam.lower_left_corner) == pytest.approx([278 + hh * 10, 278 - hh * 10, -790])
    assert cam.lens_radius == 0.0


def test_error_paths():
    with Scene(1) as sc:
        m = S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5))
        sphere = S.Sphere.new(S.vec3(0, 0, 0), 1.0, m)
        with pytest.raises(RtError) as e:
            S.FlipNormals.new_xy(sphere)  # hitable.rs:320-325: rects only
        assert e.value.code == _lib.RT_ERR_INVALID
        with pytest.raises(RtError):
            S.BvhNode.new([])
        # the one nesting the flattener does not express: a ConstantMedium inside another hitable
        fog = S.ConstantMedium.new(sphere, 0.5, S.Texture.new_constant(S.vec3(1, 1, 1)))
        with pytest.raises(RtError) as e:
            sc.flatten(S.HitableList.new([S.RotateY.new(fog, 10.0)]))
        assert e.value.code == _lib.RT_ERR_UNSUPPORTED
        assert b"ConstantMedium" in _lib.lib().rt_last_error()
        with pytest.raises(RtError):
            sc.flatten(S.HitableList.new([S.Translate.new(S.HitableList.new([]), S.vec3(1, 0, 0))]))


def test_flatten_writes_any_nesting_as_objects_in_objects(orc):
    """RotateY(Translate(x)) is a rotate-only object around a sub-tree of one leaf (an infinite box: the source makes no box
    test between the two) that holds the translate-only object around x; a HitableList of wrapped things is a sub-tree with
    one leaf per item.  The f64 and fp32 oracles walk the result to the same primitives."""
    rng = np.random.RandomState(2)
    with Scene(3) as sc:
        m = S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5))
        a = S.RotateY.new(S.Translate.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(1, 2, 3), m), S.vec3(4.0, 0.0, 0.0)), 30.0)
        b = S.Translate.new(S.HitableList.new([S.RotateY.new(S.Sphere.new(S.vec3(2, 0, 0), 1.0, m), 45.0), S.Sphere.new(S.vec3(0, 3, 0), 1.0, m)]),
                            S.vec3(-5.0, 0.0, 0.0))
        flat = sc.flatten(S.HitableList.new([a, b]))
        obs = [flat.objects[i] for i in range(flat.n_objects)]
        assert [o.kind for o in obs].count(_lib.RT_OBJ_BVH) == 2
        outer_a = [o for o in obs if o.kind == _lib.RT_OBJ_BVH and o.xform == _lib.RT_XFORM_ROTATE_Y][0]
        root = flat.nodes[outer_a.first]
        assert np.isinf(root.bmin[0]) and np.isinf(root.bmax[2]) and root.left < 0 and root.right == 0
        inner = flat.objects[~root.left]
        assert inner.kind == _lib.RT_OBJ_LIST and inner.xform == _lib.RT_XFORM_TRANSLATE and inner.count == 6
        outer_b = [o for o in obs if o.kind == _lib.RT_OBJ_BVH and o.xform == _lib.RT_XFORM_TRANSLATE][0]
        node = flat.nodes[outer_b.first]
        assert node.left >= 0 and node.right >= 0  # two items: an interior node over two leaves
        o = rng.uniform(-12, 12, size=(4000, 3))
        d = rng.uniform(-6, 6, size=(4000, 3)) - o
        p32, t32 = orc.trace_closest(flat, o.astype(np.float32), d.astype(np.float32), mode=orc.ORC_F32_FUSED)
        p64, t64 = orc.trace_closest(flat, o.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64), mode=orc.ORC_F64)
        assert (p32 >= 0).mean() > 0.05 and (p32 == p64).mean() > 0.999
        hit = (p32 >= 0) & (p32 == p64)
        assert np.allclose(t32[hit], t64[hit], rtol=1e-4)
        assert set(np.unique(p32)) >= set(range(8))  # the cuboid's six rects and both spheres are hit


def test_perlin_tables_follow_the_degenerate_shuffle():
    # perlin.rs:81-91: the swap target is always 0, so every permutation is the same fixed rotation
    with Scene(5) as sc:
        noise = S.Texture.new_noise(4.0)
        m = S.Lambertian.new(noise)
        flat = sc.flatten(S.HitableList.new([S.Sphere.new(S.vec3(0, 0, 0), 1.0, m)]))
        assert flat.n_perlins == 1
        p = flat.perlins[0]
        px, py, pz = list(p.perm_x), list(p.perm_y), list(p.perm_z)
        assert px == py == pz
        assert sorted(px) == list(range(256))
        assert px == [(k + 1) & 255 for k in range(256)]
        v = np.array([list(p.ranvec[i]) for i in range(256)])
        assert np.allclose((v ** 2).sum(axis=1), 1.0, atol=1e-6)


def test_png_round_trip(tmp_path):
    """write_png writes what read_png (and any PNG reader) reads back; the file
    layout is checked against a second, independent decoder when one is there."""
    from raytracing_amd.png import read_png, write_png
    rng = np.random.RandomState(0)
    for shape in ((1, 1, 3), (7, 13, 3), (64, 40, 3)):
        img = rng.randint(0, 256, size=shape).astype(np.uint8)
        path = str(tmp_path / ("t%d.png" % shape[0]))
        write_png(path, img)
        assert np.array_equal(read_png(path), img)
        try:
            from PIL import Image
        except ImportError:
            continue
        assert np.array_equal(np.asarray(Image.open(path).convert("RGB")), img)
    with pytest.raises(ValueError):
        write_png(str(tmp_path / "bad.png"), np.zeros((4, 4), dtype=np.uint8))


def test_light_shape_read_from_the_scene_equals_the_hard_coded_one():
    """color() hard-codes XZRect(213, 343, 227, 332, 554) (main.rs:178); the scene
    builder finds the same rect in cornell_box_scene (scenes.rs:310-317)."""
    from raytracing_amd.scenes import MAIN_RS_LIGHT_SHAPE
    with Scene(1) as a, Scene(1) as b:
        with a:
            cam, world = cornell_box_scene(1.0)
            a.light_xz_rect(*MAIN_RS_LIGHT_SHAPE)
            fa = a.flatten(world)
            typed = [fa.lights[0].f[k] for k in range(5)]
        with b:
            cam, world = cornell_box_scene(1.0)
            assert b.light_from_world(world) == 1
            fb = b.flatten(world)
            assert fb.n_lights == 1 and fb.lights[0].type == 0  # RT_LIGHT_XZ_RECT
            assert [fb.lights[0].f[k] for k in range(5)] == typed
    with Scene(2) as c:
        world = S.HitableList.new([S.Sphere.new(S.vec3(0, 0, 0), 1.0, S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5)))])
        assert c.light_from_world(world) == 0
        assert c.flatten(world).n_lights == 0


def test_registering_a_light_shape_again_replaces_it_and_flat_scenes_keep_their_owner():
    """color() samples one shape (main.rs:177-178): a second registration on the
    same Scene replaces the first (rendering twice from one Scene used to leave two
    lights, which the upload refuses); the rt_scene a flatten returns keeps its
    Scene alive, since its pointers lead into the Scene's buffers."""
    import gc
    with Scene(1) as sc:
        cam, world = cornell_box_scene(1.0)
        sc.light_xz_rect(0.0, 1.0, 0.0, 1.0, 5.0)
        sc.light_xz_rect(*MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(world)
    assert flat.n_lights == 1
    assert [flat.lights[0].f[k] for k in range(5)] == list(MAIN_RS_LIGHT_SHAPE)
    assert flat._keep is sc
    del sc, cam, world
    gc.collect()
    assert flat.n_prims == 18 and flat.prims[17].type & 0xFF in (1, 2, 3)


def test_cpp_host_program_fails_loudly_without_a_device(tmp_path):
    """There is no CPU path: on a machine without a HIP device the C++ program
    stops with the library's error, not with an image."""
    import subprocess
    exe = os.path.join(ROOT, "raytracing_amd", "csrc", "cornell_main")
    if not os.path.exists(exe):
        pytest.skip("cornell_main not built")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a device is present")
    except ImportError:
        pass
    out = str(tmp_path / "x.png")
    run = subprocess.run([exe, "1", "1", out], capture_output=True, text=True, timeout=120)
    assert run.returncode != 0
    assert "no HIP device" in run.stderr
    assert not os.path.exists(out)
    assert subprocess.run([exe, "0"], capture_output=True, text=True, timeout=60).returncode == 2


def test_rect_boxes_keep_their_thickness_at_large_coordinates():
    """A rect's box is padded by 0.0001 (hitable.rs:185-190); in fp32 that is below
    half a unit in the last place from |k| = 2048 on.  The f64 box is rounded
    outwards on its way to the fp32 node array, so the box never collapses and the
    tree still finds the rect (BVH walk = list walk, fp32 and f64)."""
    from oracle import binding as orc
    ks = [554.0, 2048.0, 3000.0, 1.0e5, -7777.0]
    with Scene(9) as sc:
        white = S.Lambertian.new_from_color(S.vec3(0.7, 0.7, 0.7))
        rects = [S.XZRect.new(-50.0, 50.0, -50.0, 50.0, k, white) for k in ks]
        rects += [S.XYRect.new(-50.0, 50.0, -50.0, 50.0, 4096.5, white), S.YZRect.new(-50.0, 50.0, -50.0, 50.0, -65536.25, white)]
        flat = sc.flatten(S.HitableList.new([S.BvhNode.new(rects)]))
    for i in range(flat.n_nodes):
        n = flat.nodes[i]
        assert all(n.bmax[c] > n.bmin[c] for c in range(3)), i
    o = np.zeros((len(ks) + 2, 3), dtype=np.float32)
    d = np.zeros_like(o)
    for i, k in enumerate(ks):
        o[i] = (3.0, k - 10.0 * np.sign(k), -4.0)
        d[i] = (0.01, np.sign(k), 0.02)
    o[-2] = (1.0, 2.0, 4000.0); d[-2] = (0.0, 0.01, 1.0)
    o[-1] = (-65000.0, 2.0, 1.0); d[-1] = (-1.0, 0.01, 0.0)
    for mode in (orc.ORC_F64, orc.ORC_F32):
        pb, tb = orc.trace_closest(flat, o, d, mode=mode, traversal=orc.ORC_TRAVERSE_BVH)
        pl, tl = orc.trace_closest(flat, o, d, mode=mode, traversal=orc.ORC_TRAVERSE_LIST)
        assert (pl >= 0).all()
        assert np.array_equal(pb, pl) and np.array_equal(tb, tl)


def _scaled_cornell(sc, k):
    """The Cornell box of scenes.rs:276-360 with every length multiplied by k."""
    red = S.Lambertian.new_from_color(S.vec3(0.65, 0.05, 0.05))
    white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
    green = S.Lambertian.new_from_color(S.vec3(0.12, 0.45, 0.15))
    light = S.DiffuseLight.new(S.Texture.new_constant(S.scalar(15.0)))
    b = 555.0 * k
    items = [S.FlipNormals.new_yz(S.YZRect.new(0.0, b, 0.0, b, b, green)), S.YZRect.new(0.0, b, 0.0, b, 0.0, red),
             S.FlipNormals.new_xz(S.XZRect.new(213 * k, 343 * k, 227 * k, 332 * k, 554 * k, light)),
             S.FlipNormals.new_xz(S.XZRect.new(0.0, b, 0.0, b, b, white)), S.XZRect.new(0.0, b, 0.0, b, 0.0, white),
             S.FlipNormals.new_xy(S.XYRect.new(0.0, b, 0.0, b, b, white)),
             S.Translate.new(S.RotateY.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(165 * k, 165 * k, 165 * k), white), -18.0),
                             S.vec3(130 * k, 0.0, 65 * k)),
             S.Translate.new(S.RotateY.new(S.Cuboid.new(S.vec3(0, 0, 0), S.vec3(165 * k, 330 * k, 165 * k), white), 15.0),
                             S.vec3(265 * k, 0.0, 295 * k))]
    cam = S.Camera.new(S.vec3(278 * k, 278 * k, -800 * k), S.vec3(278 * k, 278 * k, 0.0), S.vec3(0, 1, 0), 40.0, 1.0, 0.0, 10.0)
    sc.light_xz_rect(213 * k, 343 * k, 227 * k, 332 * k, 554 * k)
    return cam, S.HitableList.new(items)


def test_fp32_tree_walk_sees_thin_boxes_from_far_origins():
    """A rect's box is 0.0002 thick.  In fp32, seen from a few thousand units away,
    both faces give the same entry and exit distance and bvh.rs:58's `<=` would
    reject the box (a room ten times the Cornell box lost 8 % of its hits and a
    third of its light); the fp32 modes widen the exit distance by two units in the
    last place.  The fp32 tree walk then finds what the fp32 list walk and the f64
    tree walk find."""
    from oracle import binding as orc
    with Scene(3) as sc:
        cam, world = _scaled_cornell(sc, 10.0)
        flat = sc.flatten(world)
    rng = np.random.RandomState(1)
    n = 20000
    o = rng.uniform(500.0, 5000.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    for mode in (orc.ORC_F32, orc.ORC_F32_FUSED):
        pb, tb = orc.trace_closest(flat, o, d, mode=mode, traversal=orc.ORC_TRAVERSE_BVH)
        pl, tl = orc.trace_closest(flat, o, d, mode=mode, traversal=orc.ORC_TRAVERSE_LIST)
        assert (pb != pl).mean() < 2e-3, (pb != pl).mean()  # what is left: the cuboids' top-face band, as in f64
    p64, _ = orc.trace_closest(flat, o, d, mode=orc.ORC_F64)
    p32, _ = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED)
    assert (p32 != p64).mean() < 2e-3, (p32 != p64).mean()


def test_rust_declarations_in_the_integration_notes_follow_the_header():
    """The `#[repr(C)]` structs a maintainer would paste from the integration notes list the
    fields of include/rt_abi.h's structs in the same order with the same types, and
    every `pub fn` there is a symbol the header declares."""
    header = open(os.path.join(ROOT, "include", "rt_abi.h")).read() + open(os.path.join(ROOT, "include", "rt_scene_api.h")).read()
    notes = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    header_nc = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    ctype = {"float": "f32", "double": "f64", "int32_t": "i32", "uint64_t": "u64", "int": "i32"}

    def c_fields(name):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header_nc, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"const (\w+)\* (\w+)$", decl)
            if m:
                out.append((m.group(2), "*const " + m.group(1)))
                continue
            ty, rest = decl.split(" ", 1)
            for item in rest.split(","):
                item = item.strip()
                dims = re.findall(r"\[(\d+)\]", item)
                t = ctype[ty]
                for dim in reversed(dims):
                    t = "[%s; %s]" % (t, dim)
                out.append((re.sub(r"\[.*", "", item), t))
        return out

    def rust_fields(name):
        body = re.search(r"pub struct %s\s*\{(.*?)\n?\}" % name, notes, re.S).group(1)
        out = []
        for m in re.finditer(r"pub (\w+): ((?:\[+[^,]*?\]+)|(?:\*const \w+)|\w+)\s*(?:,|$)", body.replace("\n", " ")):
            out.append((m.group(1).rstrip("_") if m.group(1) == "type_" else m.group(1), m.group(2)))
        return out

    for name in ("rt_node", "rt_prim", "rt_object", "rt_material", "rt_texture", "rt_perlin", "rt_light", "rt_scene", "rt_camera",
                 "rt_render_params", "rt_stats"):
        assert rust_fields(name) == c_fields(name), (name, rust_fields(name), c_fields(name))
    for fn in re.findall(r"pub fn (\w+)\(", notes):
        assert re.search(r"\b%s\(" % fn, header_nc), fn
