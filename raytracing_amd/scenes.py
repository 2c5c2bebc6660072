"""Scene builders with the reference's signatures:
`fn *_scene(ratio: f64) -> (Camera, HitableList)` (src/scenes.rs:276).

Call them inside `with Scene(seed):`.
"""
from .scene import (Camera, Cuboid, DiffuseLight, FlipNormals, HitableList, Lambertian, RotateY, Texture,
                    Translate, XYRect, XZRect, YZRect, scalar, vec3)

# The light shape color() hard-codes for HitablePDF (src/main.rs:177-178).
MAIN_RS_LIGHT_SHAPE = (213.0, 343.0, 227.0, 332.0, 554.0)


def cornell_box_scene(ratio):
    """src/scenes.rs:276-360, value for value (the left wall's z range is 565
    there, not 555)."""
    red = Lambertian.new(Texture.new_constant(vec3(0.65, 0.05, 0.05)))
    white = Lambertian.new(Texture.new_constant(vec3(0.73, 0.73, 0.73)))
    green = Lambertian.new(Texture.new_constant(vec3(0.12, 0.45, 0.15)))
    light = DiffuseLight.new(Texture.new_constant(scalar(15.0)))

    items = [
        # Left wall
        FlipNormals.new_yz(YZRect.new(0.0, 555.0, 0.0, 565.0, 555.0, green)),
        # Right wall
        YZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, red),
        # Light
        FlipNormals.new_xz(XZRect.new(213.0, 343.0, 227.0, 332.0, 554.0, light)),
        # Ceiling
        FlipNormals.new_xz(XZRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)),
        # Floor
        XZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, white),
        # Back wall
        FlipNormals.new_xy(XYRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)),
        # Cuboids
        Translate.new(RotateY.new(Cuboid.new(scalar(0.0), scalar(165.0), white), -18.0), vec3(130.0, 0.0, 65.0)),
        Translate.new(RotateY.new(Cuboid.new(scalar(0.0), vec3(165.0, 330.0, 165.0), white), 15.0),
                      vec3(265.0, 0.0, 295.0)),
    ]
    world = HitableList.new(items)

    lookfrom = vec3(278.0, 278.0, -800.0)
    lookat = vec3(278.0, 278.0, 0.0)
    dist_to_focus = 10.0
    aperture = 0.0
    vfov = 40.0
    camera = Camera.new(lookfrom, lookat, vec3(0.0, 1.0, 0.0), vfov, ratio, aperture, dist_to_focus)
    return camera, world


# ---------------------------------------------------------------------------
# Builders that are commented out in the reference at this commit, restated
# from that text, and builders for configurations the reference has no code for.
# They draw their construction-time random numbers from a seeded
# random.Random (the reference uses the unseeded rand::thread_rng()).
# ---------------------------------------------------------------------------
import random as _random

from .scene import BvhNode, ConstantMedium, Dielectric, Isotropic, Metal, Sphere


def random_scene(ratio, seed=0x5EED0000):
    """src/scenes.rs:12-124 (commented out there): the Book-1 cover scene as this
    repository had it: a 1000-radius Lambertian ground, a 22x22 grid of small
    spheres (70 % Lambertian, 15 % metal, 5 % glass, 10 % emissive; note the
    `a + 0.9 + rand` x offset at :32), and four big spheres of which one is a
    50.0 emitter."""
    rng = _random.Random(seed)
    items = [Sphere.new(vec3(0.0, -1000.0, 0.0), 1000.0, Lambertian.new(Texture.new_constant(vec3(0.5, 0.4, 0.6))))]
    for a in range(-11, 11):
        for b in range(-11, 11):
            choose_mat = rng.random()
            center = vec3(a + 0.9 + rng.random(), 0.2, b + 0.9 * rng.random())
            d = center - vec3(4.0, 0.2, 0.0)
            if (d.x * d.x + d.y * d.y + d.z * d.z) ** 0.5 > 0.9:
                if choose_mat < 0.70:
                    m = Lambertian.new_from_color(vec3(rng.random() * rng.random(), rng.random() * rng.random(),
                                                       rng.random() * rng.random()))
                elif choose_mat < 0.85:
                    m = Metal.new(vec3(0.5 * (1.0 + rng.random()), 0.5 * (1.0 + rng.random()), 0.5 * (1.0 + rng.random())),
                                  0.5 * rng.random())
                elif choose_mat < 0.90:
                    m = Dielectric.new(1.5)
                else:
                    m = DiffuseLight.new(Texture.new_constant(vec3(4.0 * rng.random(), 4.0 * rng.random(), 4.0 * rng.random())))
                items.append(Sphere.new(center, 0.2, m))
    items.append(Sphere.new(vec3(0.0, 1.0, 0.0), 1.0, Dielectric.new(1.5)))
    items.append(Sphere.new(vec3(0.0, 3.0, 0.0), 1.0, DiffuseLight.new(Texture.new_constant(scalar(50.0)))))
    items.append(Sphere.new(vec3(-4.0, 1.0, 0.0), 1.0, Lambertian.new_from_color(vec3(0.4, 0.2, 0.1))))
    items.append(Sphere.new(vec3(4.0, 1.0, 0.0), 1.0, Metal.new(vec3(0.7, 0.6, 0.5), 0.0)))
    world = HitableList.new(items)
    camera = Camera.new(vec3(13.0, 2.0, 3.0), vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), 20.0, ratio, 0.1, 10.0)
    return camera, world


def two_spheres_scene(ratio):
    """src/scenes.rs:127-158 (commented out): two checkered spheres."""
    checker = Lambertian.new(Texture.new_checker(Texture.new_constant(vec3(0.2, 0.3, 0.1)),
                                                 Texture.new_constant(vec3(0.9, 0.9, 0.9))))
    world = HitableList.new([Sphere.new(vec3(0.0, -10.0, 0.0), 10.0, checker),
                             Sphere.new(vec3(0.0, 10.0, 0.0), 10.0, checker)])
    camera = Camera.new(vec3(13.0, 2.0, 3.0), vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), 20.0, ratio, 0.0, 10.0)
    return camera, world


def simple_light_scene(ratio):
    """src/scenes.rs:227-273 (commented out): two marble spheres and a rect light."""
    pertext = Texture.new_noise(4.0)
    world = HitableList.new([
        Sphere.new(vec3(0.0, -1000.0, 0.0), 1000.0, Lambertian.new(pertext)),
        Sphere.new(vec3(0.0, 2.0, 0.0), 2.0, Lambertian.new(pertext)),
        XYRect.new(3.0, 5.0, 1.0, 3.0, -2.0, DiffuseLight.new(Texture.new_constant(scalar(4.0)))),
    ])
    camera = Camera.new(vec3(30.0, 2.0, 3.0), vec3(0.0, 0.0, 0.0), vec3(0.0, 1.0, 0.0), 20.0, ratio, 0.0, 10.0)
    return camera, world


def cornell_smoke_scene(ratio, smoke=True, flip_light=True):
    """src/scenes.rs:363-446 (commented out): the Cornell box with the larger,
    dimmer light, a glass sphere with a mirror core, and, when `smoke` is set, the
    ConstantMedium the commented lines at :413-418 describe (density 0.01, grey,
    in a 150-radius sphere at the position of b2, :370).  The commented source has
    the light un-flipped (:392); it predates DiffuseLight emitting only towards the
    side its normal points to (material.rs:197-203), under which an un-flipped
    ceiling light shines upwards and the box stays black.  flip_light=False gives
    the scene as written, the default turns the light to face down as the live
    cornell_box_scene does (:290-297)."""
    red = Lambertian.new(Texture.new_constant(vec3(0.65, 0.05, 0.05)))
    white = Lambertian.new(Texture.new_constant(vec3(0.73, 0.73, 0.73)))
    green = Lambertian.new(Texture.new_constant(vec3(0.12, 0.45, 0.15)))
    light = DiffuseLight.new(Texture.new_constant(scalar(7.0)))
    items = [
        FlipNormals.new_yz(YZRect.new(0.0, 555.0, 0.0, 565.0, 555.0, green)),
        YZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, red),
        FlipNormals.new_xz(XZRect.new(113.0, 443.0, 127.0, 432.0, 554.0, light)) if flip_light else XZRect.new(113.0, 443.0, 127.0, 432.0, 554.0, light),
        FlipNormals.new_xz(XZRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)),
        XZRect.new(0.0, 555.0, 0.0, 555.0, 0.0, white),
        FlipNormals.new_xy(XYRect.new(0.0, 555.0, 0.0, 555.0, 555.0, white)),
    ]
    if smoke:
        b2 = Sphere.new(vec3(265.0, 150.0, 295.0), 150.0, white)
        items.append(ConstantMedium.new(b2, 0.01, Texture.new_constant(scalar(0.5))))
    items.append(Sphere.new(vec3(277.5, 150.0, 295.0), 120.0, Dielectric.new(1.5)))
    items.append(Sphere.new(vec3(277.5, 150.0, 295.0), 118.0, Metal.new(scalar(1.0), 0.0)))
    world = HitableList.new(items)
    camera = Camera.new(vec3(278.0, 278.0, -800.0), vec3(278.0, 278.0, 0.0), vec3(0.0, 1.0, 0.0), 40.0, ratio, 0.0, 10.0)
    return camera, world


# The light shape a caller registers for cornell_smoke_scene (its XZRect light, scenes.rs:392).
SMOKE_LIGHT_SHAPE = (113.0, 443.0, 127.0, 432.0, 554.0)


def final_scene(ratio, seed=0x5EED0002, boxes_per_side=20, cluster=1000, marble=True, media=True):
    """The closing scene of "Ray Tracing: The Next Week", which the reference's
    README lists as not reached (README.md:12-14): it has no code for it, so this
    builder is written from the book with the pieces the reference does have: a
    BvhNode over boxes_per_side^2 ground boxes of random height, a rect light, a
    glass sphere, a metal sphere, two constant media (a blue one inside a glass
    sphere and a thin mist around everything), a Perlin marble sphere, and a
    BvhNode of `cluster` small white spheres under RotateY + Translate.  No motion
    blur and no image texture: the reference has neither."""
    rng = _random.Random(seed)
    ground = Lambertian.new_from_color(vec3(0.48, 0.83, 0.53))
    boxes = []
    w = 100.0
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            x0 = -1000.0 + i * w
            z0 = -1000.0 + j * w
            y1 = 100.0 * (rng.random() + 0.01)
            boxes.append(Cuboid.new(vec3(x0, 0.0, z0), vec3(x0 + w, y1, z0 + w), ground))
    items = [BvhNode.new(boxes)]
    # DiffuseLight emits towards the side its normal points to (material.rs:197-203), so a ceiling light is flipped to face
    # down, as the reference's Cornell box does (scenes.rs:290-297)
    items.append(FlipNormals.new_xz(XZRect.new(123.0, 423.0, 147.0, 412.0, 554.0, DiffuseLight.new(Texture.new_constant(scalar(7.0))))))
    items.append(Sphere.new(vec3(400.0, 400.0, 200.0), 50.0, Lambertian.new_from_color(vec3(0.7, 0.3, 0.1))))
    items.append(Sphere.new(vec3(260.0, 150.0, 45.0), 50.0, Dielectric.new(1.5)))
    items.append(Sphere.new(vec3(0.0, 150.0, 145.0), 50.0, Metal.new(vec3(0.8, 0.8, 0.9), 10.0)))
    items.append(Sphere.new(vec3(360.0, 150.0, 145.0), 70.0, Dielectric.new(1.5)))
    if media:  # (the switches exist for timing experiments: what the media and the marble cost)
        items.append(ConstantMedium.new(Sphere.new(vec3(360.0, 150.0, 145.0), 70.0, Dielectric.new(1.5)), 0.2,
                                        Texture.new_constant(vec3(0.2, 0.4, 0.9))))
        items.append(ConstantMedium.new(Sphere.new(vec3(0.0, 0.0, 0.0), 5000.0, Dielectric.new(1.5)), 0.0001,
                                        Texture.new_constant(scalar(1.0))))
    items.append(Sphere.new(vec3(220.0, 280.0, 300.0), 80.0,
                            Lambertian.new(Texture.new_noise(0.1)) if marble else Lambertian.new_from_color(vec3(0.5, 0.5, 0.5))))
    white = Lambertian.new_from_color(vec3(0.73, 0.73, 0.73))
    spheres = [Sphere.new(vec3(165.0 * rng.random(), 165.0 * rng.random(), 165.0 * rng.random()), 10.0, white)
               for _ in range(cluster)]
    items.append(Translate.new(RotateY.new(BvhNode.new(spheres), 15.0), vec3(-100.0, 270.0, 395.0)))
    world = HitableList.new(items)
    camera = Camera.new(vec3(478.0, 278.0, -600.0), vec3(278.0, 278.0, 0.0), vec3(0.0, 1.0, 0.0), 40.0, ratio, 0.0, 10.0)
    return camera, world


FINAL_LIGHT_SHAPE = (123.0, 423.0, 147.0, 412.0, 554.0)


def sphere_stress_scene(ratio, n_spheres=1 << 20, seed=0x5EED0003):
    """Traversal stress scene (no counterpart in the reference): n_spheres
    Lambertian spheres with centres uniform in a 1000-unit cube and radii small
    enough that they rarely overlap, under one large rect light high above."""
    rng = _random.Random(seed)
    side = 1000.0
    r = 0.25 * side / (n_spheres ** (1.0 / 3.0))
    mats = [Lambertian.new_from_color(vec3(0.2 + 0.6 * rng.random(), 0.2 + 0.6 * rng.random(), 0.2 + 0.6 * rng.random()))
            for _ in range(16)]
    items = [Sphere.new(vec3(side * rng.random(), side * rng.random(), side * rng.random()), r * (0.5 + rng.random()),
                        mats[k & 15]) for k in range(n_spheres)]
    items.append(FlipNormals.new_xz(XZRect.new(-500.0, 1500.0, -500.0, 1500.0, 3000.0, DiffuseLight.new(Texture.new_constant(scalar(10.0))))))  # faces down
    world = HitableList.new(items)
    # from above the front edge, so the view is of the faces the light reaches (the cloud is opaque a few layers in)
    camera = Camera.new(vec3(500.0, 1900.0, -1300.0), vec3(500.0, 500.0, 500.0), vec3(0.0, 1.0, 0.0), 40.0, ratio, 0.0, 10.0)
    return camera, world


STRESS_LIGHT_SHAPE = (-500.0, 1500.0, -500.0, 1500.0, 3000.0)
