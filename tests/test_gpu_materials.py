"""Parity of the materials, textures and media that are commented out in the
reference at this commit (Metal, Dielectric, Isotropic + ConstantMedium,
checker and Perlin noise textures, the sky gradient, the thin-lens camera):
HIP render against the CPU oracle's restatement of that commented-out text.
Parity here is pinned by the restatement only (the reference has no output
that exercises these paths)."""
import numpy as np
import pytest

from raytracing_amd import Scene, _lib, make_params
from raytracing_amd import scene as S
from raytracing_amd import scenes
from conftest import log_achieved
from test_gpu_render import _z_scores

pytestmark = pytest.mark.gpu


def _compare(gpu, orc, flat, camera, nx, ny, spp, sky=_lib.RT_SKY_BLACK, max_depth=50, same_frac=0.985, tight_frac=0.95, energy_tol=0.005):
    """Same seed against the fp32 oracle, then independent seeds against the f64 oracle (3-standard-error test).
    Same seed: the device draws the oracle's random numbers, so a pixel sum differs only where a last-bit difference
    in shading sent a path to another surface: at least `same_frac` of the sums agree to 0.5 %, at least `tight_frac`
    to fp32 accuracy (2e-4), and the image's energy to `energy_tol`.  The bars are what the kernels achieve on these
    scenes (0.997 or more of the sums to 0.5 %, 0.977 or more to 2e-4, energy to 2e-4) less a margin; the room ten times
    the Cornell box, where fp32 coordinates are coarser, gets 0.988 / 0.911 and has bars of its own."""
    gpu.upload_scene(flat)
    p = make_params(nx, ny, spp, seed=77, sky=sky, max_depth=max_depth)
    s, s2 = gpu.render(camera, p, want_sumsq=True)
    o, o2 = orc.render(flat, camera.c, p, mode=orc.ORC_F32_FUSED)
    assert np.isfinite(s).all()
    close = np.isclose(s, o, rtol=5e-3, atol=5e-3 * spp).mean()
    tight = np.isclose(s, o, rtol=2e-4, atol=1e-4 * spp).mean()
    log_achieved("same_seed", close=close, tight=tight, energy=abs(s.sum() - o.sum()) / o.sum(), bar=same_frac)
    assert close > same_frac, close
    assert tight > tight_frac, tight
    assert abs(s.sum() - o.sum()) / o.sum() < energy_tol, (s.sum(), o.sum())
    q = make_params(nx, ny, spp, seed=78, sky=sky, max_depth=max_depth)
    r, r2 = orc.render(flat, camera.c, q, mode=orc.ORC_F64)
    ma, mb, se, z = _z_scores(s.astype(np.float64), s2.astype(np.float64), r, r2, spp)
    valid = se > 0
    zz = z[valid]
    log_achieved("z_test", frac_over_3=(np.abs(zz) > 3).mean(), mean_z=zz.mean())
    assert (np.abs(zz) > 3).mean() < 0.02, (np.abs(zz) > 3).mean()
    assert abs(zz.mean()) < 0.08, zz.mean()
    # image means agree within 4 standard errors of their difference (or 3 %)
    se_mean = np.sqrt((se ** 2).sum()) / se.size
    assert abs(ma.mean() - mb.mean()) < max(4.0 * se_mean, 0.03 * mb.mean()), (ma.mean(), mb.mean(), se_mean)


def test_metal_dielectric_lens_and_sky_gradient(gpu, orc):
    """A small Book-1 style scene: Lambertian ground, metal (fuzzy and mirror),
    glass, an emitter, thin-lens camera (aperture 0.1) and the gradient sky."""
    with Scene(5) as sc:
        items = [
            S.Sphere.new(S.vec3(0.0, -1000.0, 0.0), 1000.0, S.Lambertian.new_from_color(S.vec3(0.5, 0.4, 0.6))),
            S.Sphere.new(S.vec3(0.0, 1.0, 0.0), 1.0, S.Dielectric.new(1.5)),
            S.Sphere.new(S.vec3(-4.0, 1.0, 0.0), 1.0, S.Lambertian.new_from_color(S.vec3(0.4, 0.2, 0.1))),
            S.Sphere.new(S.vec3(4.0, 1.0, 0.0), 1.0, S.Metal.new(S.vec3(0.7, 0.6, 0.5), 0.0)),
            S.Sphere.new(S.vec3(2.0, 0.5, 2.0), 0.5, S.Metal.new(S.vec3(0.8, 0.8, 0.3), 0.4)),
            S.Sphere.new(S.vec3(-1.5, 0.4, 2.5), 0.4, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(4.0)))),
        ]
        cam = S.Camera.new(S.vec3(13.0, 2.0, 3.0), S.vec3(0.0, 0.0, 0.0), S.vec3(0.0, 1.0, 0.0), 20.0, 1.5, 0.1, 10.0)
        flat = sc.flatten(S.HitableList.new(items))
        _compare(gpu, orc, flat, cam, 60, 40, 64, sky=_lib.RT_SKY_GRADIENT)


def test_checker_and_noise_textures(gpu, orc):
    """two_spheres_scene's checker and simple_light_scene's marble, lit by a rect
    light and the gradient sky."""
    with Scene(6) as sc:
        checker = S.Lambertian.new(S.Texture.new_checker(S.Texture.new_constant(S.vec3(0.2, 0.3, 0.1)),
                                                         S.Texture.new_constant(S.vec3(0.9, 0.9, 0.9))))
        marble = S.Lambertian.new(S.Texture.new_noise(4.0))
        items = [
            S.Sphere.new(S.vec3(0.0, -1000.0, 0.0), 1000.0, checker),
            S.Sphere.new(S.vec3(0.0, 2.0, 0.0), 2.0, marble),
            S.XYRect.new(3.0, 5.0, 1.0, 3.0, -2.0, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(4.0)))),
        ]
        cam = S.Camera.new(S.vec3(30.0, 2.0, 3.0), S.vec3(0.0, 0.0, 0.0), S.vec3(0.0, 1.0, 0.0), 20.0, 1.5, 0.0, 10.0)
        flat = sc.flatten(S.HitableList.new(items))
        assert flat.n_perlins == 1
        _compare(gpu, orc, flat, cam, 60, 40, 64, sky=_lib.RT_SKY_GRADIENT)


def test_cornell_smoke_with_glass_and_mirror(gpu, orc):
    """cornell_smoke_scene (scenes.rs:363-446): ConstantMedium in a sphere, a glass
    sphere with a mirror core, MixturePDF sampling of its own light."""
    with Scene(7) as sc:
        cam, world = scenes.cornell_smoke_scene(1.0, smoke=True)
        sc.light_xz_rect(*scenes.SMOKE_LIGHT_SHAPE)
        flat = sc.flatten(world)
        assert flat.n_objects == 1 and flat.objects[0].kind == _lib.RT_OBJ_MEDIUM
        _compare(gpu, orc, flat, cam, 48, 48, 64)


def test_medium_in_rotated_box(gpu, orc):
    """A ConstantMedium whose boundary is Translate(RotateY(Cuboid)), the nesting
    the commented line scenes.rs:413 uses."""
    with Scene(8) as sc:
        cam, world0 = scenes.cornell_box_scene(1.0)
        white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
        box = S.Translate.new(S.RotateY.new(S.Cuboid.new(S.scalar(0.0), S.vec3(165.0, 330.0, 165.0), white), 15.0),
                              S.vec3(265.0, 0.0, 295.0))
        items = world0.list[:6] + [S.ConstantMedium.new(box, 0.01, S.Texture.new_constant(S.scalar(0.9)))]
        sc.light_xz_rect(*scenes.MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(S.HitableList.new(items))
        _compare(gpu, orc, flat, cam, 48, 48, 64)


def test_trace_skips_media_and_sees_general_scenes(gpu, orc):
    """rt_trace_closest on a scene with a medium: deterministic, the medium is
    invisible, everything else matches the fp32 oracle bit for bit."""
    with Scene(9) as sc:
        cam, world = scenes.cornell_smoke_scene(1.0, smoke=True)
        flat = sc.flatten(world)
        gpu.upload_scene(flat)
        rng = np.random.RandomState(4)
        o = (rng.rand(20000, 3) * 555.0).astype(np.float32)
        d = rng.randn(20000, 3).astype(np.float32)
        prim, t = gpu.trace_closest(o, d)
        oprim, ot = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED)
        assert np.array_equal(prim, oprim)
        assert np.array_equal(t, ot.astype(np.float32))
        assert prim.max() < flat.n_prims


def _sphere_field(n, seed, glass_and_metal):
    """n small spheres over a big ground sphere, a rect light above them."""
    rng = np.random.RandomState(seed)
    items = [S.Sphere.new(S.vec3(0.0, -1000.0, 0.0), 1000.0, S.Lambertian.new_from_color(S.vec3(0.5, 0.5, 0.5)))]
    for k in range(n):
        c = S.vec3(float(rng.uniform(-6, 6)), 0.3, float(rng.uniform(-6, 6)))
        pick = rng.rand() if glass_and_metal else 0.0
        if pick < 0.7:
            m = S.Lambertian.new_from_color(S.vec3(*(rng.rand(3) * rng.rand(3))))
        elif pick < 0.85:
            m = S.Metal.new(S.vec3(*(0.5 * (1 + rng.rand(3)))), float(0.5 * rng.rand()))
        else:
            m = S.Dielectric.new(1.5)
        items.append(S.Sphere.new(c, 0.3, m))
    # facing down, like the Cornell box's lamp (scenes.rs:310-317)
    items.append(S.FlipNormals.new_xz(S.XZRect.new(-3.0, 3.0, -3.0, 3.0, 6.0, S.DiffuseLight.new(S.Texture.new_constant(S.scalar(6.0))))))
    cam = S.Camera.new(S.vec3(10.0, 3.0, 6.0), S.vec3(0.0, 0.3, 0.0), S.vec3(0.0, 1.0, 0.0), 35.0, 1.5, 0.0, 10.0)
    return cam, S.HitableList.new(items)


def test_large_tree_render_lambertian(gpu, orc):
    """150 spheres: more leaves than a flat array takes, so the per-lane stack
    traversal runs under the Lambertian-only build of the render kernel, with the
    MixturePDF sampling the rect light."""
    with Scene(21) as sc:
        cam, world = _sphere_field(150, 5, glass_and_metal=False)
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        flat = sc.flatten(world)
        assert flat.n_nodes > 64
        _compare(gpu, orc, flat, cam, 60, 40, 64)


def test_large_tree_render_all_materials(gpu, orc):
    """The same field with metal and glass spheres and the gradient sky: stack
    traversal under the general build."""
    with Scene(22) as sc:
        cam, world = _sphere_field(150, 6, glass_and_metal=True)
        flat = sc.flatten(world)
        _compare(gpu, orc, flat, cam, 60, 40, 64, sky=_lib.RT_SKY_GRADIENT)


def test_long_primitive_list_render(gpu, orc):
    """One Translate(RotateY(HitableList)) of 80 rects: a single leaf with more
    primitives than the flat layout's LDS tables take, so the hierarchical scan
    runs it (wave-uniform loop over the list)."""
    with Scene(23) as sc:
        cam, world0 = scenes.cornell_box_scene(1.0)
        white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
        rng = np.random.RandomState(2)
        slats = []
        for k in range(80):
            x0, z0 = float(rng.uniform(0, 140)), float(rng.uniform(0, 140))
            slats.append(S.XZRect.new(x0, x0 + 25.0, z0, z0 + 25.0, float(5.0 + 4.0 * k), white))
        blinds = S.Translate.new(S.RotateY.new(S.HitableList.new(slats), 20.0), S.vec3(200.0, 0.0, 150.0))
        sc.light_xz_rect(*scenes.MAIN_RS_LIGHT_SHAPE)
        flat = sc.flatten(S.HitableList.new(world0.list[:6] + [blinds]))
        assert flat.n_prims > 64
        _compare(gpu, orc, flat, cam, 48, 48, 64)


@pytest.mark.parametrize("name", ["random", "final", "stress"])
def test_named_scene_builders_render_like_the_oracle(gpu, orc, name):
    """The builders behind the other named configurations (Book-1 random spheres,
    the Book-2 final scene with its box grid, sphere cluster, marble, glass and two
    media, and the sphere stress scene at 4096 spheres), at thumbnail size: same
    seed against the fp32 oracle, other seed against the f64 oracle."""
    ratio = 1.5
    with Scene(31) as sc:
        if name == "random":
            cam, world = scenes.random_scene(ratio)
        elif name == "final":
            cam, world = scenes.final_scene(ratio)
            sc.light_xz_rect(*scenes.FINAL_LIGHT_SHAPE)
        else:
            cam, world = scenes.sphere_stress_scene(ratio, 4096)
            sc.light_xz_rect(*scenes.STRESS_LIGHT_SHAPE)
        flat = sc.flatten(world)
        assert flat.n_nodes > 64
        _compare(gpu, orc, flat, cam, 60, 40, 64)


def test_room_ten_times_the_cornell_box(gpu, orc):
    """Every length of the Cornell box times ten (walls at 5550): the scale at which
    fp32 loses a rect's 0.0002-thick box unless the slab test allows for it.  Trace
    bit-exact against the fp32 oracle, image against the f64 one."""
    from test_scene_host import _scaled_cornell
    with Scene(41) as sc:
        cam, world = _scaled_cornell(sc, 10.0)
        flat = sc.flatten(world)
    gpu.upload_scene(flat)
    rng = np.random.RandomState(2)
    n = 30000
    o = rng.uniform(500.0, 5000.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    prim, t = gpu.trace_closest(o, d)
    op, ot = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED)
    assert np.array_equal(prim, op) and np.array_equal(t, ot.astype(np.float32))
    assert (prim >= 0).mean() > 0.85
    _compare(gpu, orc, flat, cam, 48, 48, 128, same_frac=0.975, tight_frac=0.88)


def test_medium_in_a_box_of_the_ten_times_room(gpu, orc):
    """The room of the previous test with its tall box turned into a ConstantMedium
    (density scaled down by ten): boundary distances beyond 2048, where fp32 cannot
    add hitable.rs:566's 0.0001 to the first boundary hit.  The second search starts
    one float further at least; fp32 then sees the medium as often as f64 does."""
    from test_scene_host import _scaled_cornell
    with Scene(43) as sc:
        cam, world0 = _scaled_cornell(sc, 10.0)
        white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
        box = S.Translate.new(S.RotateY.new(S.Cuboid.new(S.scalar(0.0), S.vec3(1650.0, 3300.0, 1650.0), white), 15.0),
                              S.vec3(2650.0, 0.0, 2950.0))
        items = world0.list[:6] + [S.ConstantMedium.new(box, 0.001, S.Texture.new_constant(S.scalar(0.9)))]
        flat = sc.flatten(S.HitableList.new(items))
    p = make_params(48, 48, 64, seed=5)
    _, st32 = orc.render(flat, cam.c, p, mode=orc.ORC_F32_FUSED, want_sumsq=False, want_stats=True)
    _, st64 = orc.render(flat, cam.c, p, mode=orc.ORC_F64, want_sumsq=False, want_stats=True)
    assert abs(st32.hits - st64.hits) < 0.003 * st64.hits, (st32.hits, st64.hits)
    _compare(gpu, orc, flat, cam, 48, 48, 64, same_frac=0.975, tight_frac=0.88)


def test_long_lists_as_leaves_of_a_large_tree(gpu, orc):
    """Leaves of 300 primitives, one a plain HitableList and one under
    Translate(RotateY(..)), among 150 spheres: the per-lane stack traversal walks
    such a list with a per-lane loop, whatever its length."""
    with Scene(24) as sc:
        cam, world = _sphere_field(150, 7, glass_and_metal=False)
        rng = np.random.RandomState(3)
        white = S.Lambertian.new_from_color(S.vec3(0.73, 0.73, 0.73))
        red = S.Lambertian.new_from_color(S.vec3(0.65, 0.05, 0.05))

        def slats(mat, n):
            out = []
            for k in range(n):
                x0, z0 = float(rng.uniform(-2.0, 1.5)), float(rng.uniform(-2.0, 1.5))
                out.append(S.XZRect.new(x0, x0 + 0.5, z0, z0 + 0.5, float(1.0 + 0.01 * k), mat))
            return S.HitableList.new(out)

        items = list(world.list)
        items.append(slats(white, 300))
        items.append(S.Translate.new(S.RotateY.new(slats(red, 300), 30.0), S.vec3(1.0, 0.5, -1.0)))
        sc.light_xz_rect(-3.0, 3.0, -3.0, 3.0, 6.0)
        flat = sc.flatten(S.HitableList.new(items))
        assert flat.n_prims > 700
        _compare(gpu, orc, flat, cam, 60, 40, 64)
        # and the nearest hits of rays aimed down at the slats, bit for bit
        n = 6000
        o = np.stack([rng.uniform(-2.5, 2.5, n), np.full(n, 5.5), rng.uniform(-2.5, 2.5, n)], axis=1).astype(np.float32)
        d = np.stack([rng.uniform(-0.3, 0.3, n), np.full(n, -1.0), rng.uniform(-0.3, 0.3, n)], axis=1).astype(np.float32)
        prim, t = gpu.trace_closest(o, d)
        op, ot = orc.trace_closest(flat, o, d, mode=orc.ORC_F32_FUSED)
        assert np.array_equal(prim, op) and np.array_equal(t, ot.astype(np.float32))
        assert (prim >= 150).mean() > 0.3  # a good share of them end on a slat
