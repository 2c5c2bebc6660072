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
